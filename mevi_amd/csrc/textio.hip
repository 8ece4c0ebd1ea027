// Host-side text formatting of the ranked TSVs (no device code): the dense result file of faiss_search.to_file
// (MEVI/faiss_search.py:71-77) holds nq x k ids and nq x k scores written as Python `str(int)` / `str(float)` of
// the f32 widened to double -- 14 M numbers at MS MARCO size, 5.7 s of Python string work per file against a 0.1 s
// search.  These two functions produce the same bytes (`repr(float)`: shortest digits that round-trip, fixed notation
// for 1e-4 <= |x| < 1e16 with at least ".0", else d.ddde+XX with a two-digit exponent; inf / nan / -0.0 as Python).
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// one double -> Python repr; returns the end pointer (buffer has >= 32 bytes left)
char *py_repr(double x, char *p) {
  if (std::isnan(x)) { std::memcpy(p, "nan", 3); return p + 3; }
  if (std::isinf(x)) {
    if (x < 0) *p++ = '-';
    std::memcpy(p, "inf", 3);
    return p + 3;
  }
  if (std::signbit(x)) { *p++ = '-'; x = -x; }
  if (x == 0.0) { std::memcpy(p, "0.0", 3); return p + 3; }
  char sci[40];
  const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);  // d[.ddd]e[+-]XX, shortest
  const char *e = sci;
  while (*e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (const char *c = sci; c < e; ++c)
    if (*c != '.') digits[nd++] = *c;
  int ex = 0;
  {
    const char *c = e + 1;
    const bool neg = *c == '-';
    ++c;
    while (c < r.ptr) ex = ex * 10 + (*c++ - '0');
    if (neg) ex = -ex;
  }
  const int decpt = ex + 1;                 // position of the decimal point relative to the digit string
  if (decpt > -4 && decpt <= 16) {          // float_repr_style 'r': fixed notation
    if (decpt <= 0) {
      *p++ = '0'; *p++ = '.';
      for (int i = 0; i < -decpt; ++i) *p++ = '0';
      std::memcpy(p, digits, nd);
      return p + nd;
    }
    if (decpt >= nd) {
      std::memcpy(p, digits, nd);
      p += nd;
      for (int i = nd; i < decpt; ++i) *p++ = '0';
      *p++ = '.'; *p++ = '0';
      return p;
    }
    std::memcpy(p, digits, decpt);
    p += decpt;
    *p++ = '.';
    std::memcpy(p, digits + decpt, nd - decpt);
    return p + (nd - decpt);
  }
  *p++ = digits[0];
  if (nd > 1) {
    *p++ = '.';
    std::memcpy(p, digits + 1, nd - 1);
    p += nd - 1;
  }
  *p++ = 'e';
  *p++ = ex < 0 ? '-' : '+';
  const int ax = ex < 0 ? -ex : ex;
  if (ax >= 100) *p++ = (char)('0' + ax / 100);
  *p++ = (char)('0' + (ax / 10) % 10);
  *p++ = (char)('0' + ax % 10);
  return p;
}

}  // namespace

// comma-joined Python reprs of the f32 values widened to double; returns the byte count (cap must be >= 26 n, else
// MEVI_ERR_INVALID_ARG)
extern "C" int64_t mevi_format_f32_list(const float *v, int64_t n, char *out, int64_t cap) {
  if (n <= 0) return 0;
  if (!v || !out) { mevi::set_error("format_f32_list: null pointer"); return MEVI_ERR_INVALID_ARG; }
  if (cap < n * 26) { mevi::set_error("format_f32_list: buffer too small"); return MEVI_ERR_INVALID_ARG; }  // sign, 17 digits, point, e+XXX, comma
  char *p = out;
  for (int64_t i = 0; i < n; ++i) {
    if (i) *p++ = ',';
    p = py_repr((double)v[i], p);
  }
  return p - out;
}

// comma-joined decimal i64
extern "C" int64_t mevi_format_i64_list(const int64_t *v, int64_t n, char *out, int64_t cap) {
  if (n <= 0) return 0;
  if (!v || !out) { mevi::set_error("format_i64_list: null pointer"); return MEVI_ERR_INVALID_ARG; }
  if (cap < n * 21) { mevi::set_error("format_i64_list: buffer too small"); return MEVI_ERR_INVALID_ARG; }
  char *p = out;
  for (int64_t i = 0; i < n; ++i) {
    if (i) *p++ = ',';
    p = std::to_chars(p, p + 21, v[i]).ptr;
  }
  return p - out;
}

// The two list columns of a whole ranked TSV (faiss_search.to_file, MEVI/faiss_search.py:71-77: 6980 x 1000 ids and scores = 14 M
// renderings per dense file) in one call: row r -> `id,id,...<TAB>score,score,...` at out + r * row_cap (row_cap >= 47 k + 1),
// its length in lens[r].  Rows are independent: `threads` host threads take contiguous row ranges (0.32 s -> ~0.06 s for the
// MS MARCO dev file).  The same bytes as the two list formatters above.
extern "C" int mevi_format_ranked_rows(const int64_t *ids, const float *scores, int64_t rows, int64_t k, char *out, int64_t row_cap,
                                       int64_t *lens, int32_t threads) {
  if (rows <= 0) return MEVI_OK;
  if (!ids || !scores || !out || !lens || k < 0) { mevi::set_error("format_ranked_rows: null pointer"); return MEVI_ERR_INVALID_ARG; }
  if (row_cap < 47 * k + 1) { mevi::set_error("format_ranked_rows: row_cap %lld < 47 k + 1", (long long)row_cap); return MEVI_ERR_INVALID_ARG; }
  auto work = [=](int64_t r0, int64_t r1) {
    for (int64_t r = r0; r < r1; ++r) {
      char *p = out + r * row_cap;
      const int64_t *iv = ids + r * k;
      const float *sv = scores + r * k;
      for (int64_t i = 0; i < k; ++i) {
        if (i) *p++ = ',';
        p = std::to_chars(p, p + 21, iv[i]).ptr;
      }
      *p++ = '\t';
      for (int64_t i = 0; i < k; ++i) {
        if (i) *p++ = ',';
        p = py_repr((double)sv[i], p);
      }
      lens[r] = p - (out + r * row_cap);
    }
  };
  int64_t nt = threads < 1 ? 1 : threads;
  if (nt > rows) nt = rows;
  if (nt == 1) {
    work(0, rows);
    return MEVI_OK;
  }
  std::vector<std::thread> pool;
  for (int64_t t = 0; t < nt; ++t) pool.emplace_back(work, rows * t / nt, rows * (t + 1) / nt);
  for (auto &th : pool) th.join();
  return MEVI_OK;
}

// ---- readers: one comma-separated TSV field -> numbers (evaluate.py / ensemble_*.py eval() every field; a dense file is
// 14 M numbers).  Return the count, or MEVI_ERR_INVALID_ARG when a token is not a plain number of that kind or `cap`
// is short -- the caller then falls back to Python's own parsing, so odd spellings keep their Python meaning.
extern "C" int64_t mevi_parse_i64_list(const char *s, int64_t len, int64_t *out, int64_t cap) {
  if (!s || !out || len <= 0) return MEVI_ERR_INVALID_ARG;
  const char *p = s, *end = s + len;
  int64_t n = 0;
  while (true) {
    while (p < end && *p == ' ') ++p;
    if (n >= cap) return MEVI_ERR_INVALID_ARG;
    const char *q = (p < end && *p == '+') ? p + 1 : p;
    const auto r = std::from_chars(q, end, out[n]);
    if (r.ec != std::errc() || r.ptr == q || (q != p && *q == '-')) return MEVI_ERR_INVALID_ARG;
    ++n;
    p = r.ptr;
    while (p < end && *p == ' ') ++p;
    if (p == end) return n;
    if (*p != ',') return MEVI_ERR_INVALID_ARG;
    ++p;
  }
}

extern "C" int64_t mevi_parse_f64_list(const char *s, int64_t len, double *out, int64_t cap) {
  if (!s || !out || len <= 0) return MEVI_ERR_INVALID_ARG;
  const char *p = s, *end = s + len;
  int64_t n = 0;
  while (true) {
    while (p < end && *p == ' ') ++p;
    if (n >= cap) return MEVI_ERR_INVALID_ARG;
    const char *q = (p < end && *p == '+') ? p + 1 : p;
    const auto r = std::from_chars(q, end, out[n], std::chars_format::general);
    if (r.ec != std::errc() || r.ptr == q || (q != p && *q == '-')) return MEVI_ERR_INVALID_ARG;
    ++n;
    p = r.ptr;
    while (p < end && *p == ' ') ++p;
    if (p == end) return n;
    if (*p != ',') return MEVI_ERR_INVALID_ARG;
    ++p;
  }
}

// A whole ranked TSV in one call (the files evaluate.py / ensemble_marco.py read: MEVI/ensemble_marco.py:92-111 `parse_file`
// eval()s one field at a time): every line's query field as a (start, length) span of `buf`, and the comma lists of up to two
// columns -- `col_i` integers, `col_f` floats (either may be -1) -- as flat arrays with per-line offsets.  Returns the number of
// lines, or MEVI_ERR_INVALID_ARG as soon as anything is not the plain shape (a missing column, an empty or bracketed
// field, a token the list parsers above refuse, a carriage return, short capacities): the caller then parses that file the
// reference's way, so odd files keep their Python meaning.
extern "C" int64_t mevi_parse_tsv_columns(const char *buf, int64_t len, int32_t col_q, int32_t col_i, int32_t col_f,
                                          int64_t *q_span, int64_t *seg_i, int64_t *vals_i, int64_t cap_i, int64_t *seg_f,
                                          double *vals_f, int64_t cap_f, int64_t cap_lines) {
  if (!buf || len < 0 || col_q < 0 || !q_span || cap_lines < 0) return MEVI_ERR_INVALID_ARG;
  if ((col_i >= 0 && (!seg_i || !vals_i)) || (col_f >= 0 && (!seg_f || !vals_f))) return MEVI_ERR_INVALID_ARG;
  const char *p = buf, *end = buf + len;
  int64_t line = 0, ni = 0, nf = 0;
  if (col_i >= 0) seg_i[0] = 0;
  if (col_f >= 0) seg_f[0] = 0;
  while (p < end) {
    const char *eol = static_cast<const char *>(std::memchr(p, '\n', end - p));
    if (!eol) eol = end;
    if (line >= cap_lines) return MEVI_ERR_INVALID_ARG;
    bool got_q = false, got_i = col_i < 0, got_f = col_f < 0;
    int col = 0;
    const char *f = p;
    while (true) {
      const char *tab = static_cast<const char *>(std::memchr(f, '\t', eol - f));
      const char *fe = tab ? tab : eol;
      if (std::memchr(f, '\r', fe - f)) return MEVI_ERR_INVALID_ARG;
      if (col == col_q) {
        q_span[2 * line] = f - buf;
        q_span[2 * line + 1] = fe - f;
        got_q = true;
      }
      if (col == col_i) {
        const int64_t n = fe > f ? mevi_parse_i64_list(f, fe - f, vals_i + ni, cap_i - ni) : -1;
        if (n < 0) return MEVI_ERR_INVALID_ARG;
        ni += n;
        got_i = true;
      }
      if (col == col_f) {
        const int64_t n = fe > f ? mevi_parse_f64_list(f, fe - f, vals_f + nf, cap_f - nf) : -1;
        if (n < 0) return MEVI_ERR_INVALID_ARG;
        nf += n;
        got_f = true;
      }
      ++col;
      if (!tab) break;
      f = tab + 1;
    }
    if (!got_q || !got_i || !got_f) return MEVI_ERR_INVALID_ARG;
    ++line;
    if (col_i >= 0) seg_i[line] = ni;
    if (col_f >= 0) seg_f[line] = nf;
    p = eol < end ? eol + 1 : end;
  }
  return line;
}
