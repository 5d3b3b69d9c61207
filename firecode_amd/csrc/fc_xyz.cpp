// fc_xyz.cpp -- the multi-conformer .xyz wire format either side of the hot path
// (SURVEY.md section 8f, rank 2).  Host code, no GPU involved.
//
//   writer, Ensemble.to_xyz (firecode/ensemble.py:284-297):
//       "{A}\nExported from FIRECODE Ensemble ({basename})\n" +
//       "\n".join("{atom} {x:15.8f} {y:15.8f} {z:15.8f}")      conformers joined by "\n",
//       no newline at the end of the file
//   writer, utils.write_xyz (firecode/utils.py:105-116):
//       "{A}\n{title}\n" + "%s     % .6f % .6f % .6f\n" per atom
//   reader, Ensemble.from_xyz (firecode/ensemble.py:58-98):
//       blank lines skipped; count line, comment line, `count` lines "sym x y z ..."
//
// Text must be identical to what Python produces, so the fixed-point formatter
// rounds the EXACT binary value to the requested decimals (as Python's format
// and glibc's printf do): the product x*10^d is split into its rounded value
// and the exact remainder with one fma, which decides every case except
// remainders within 1e-15 of a tie -- those fall back to snprintf.  Numbers are
// parsed with Clinger's exact fast path (integer mantissa < 2^53, |exp10| <= 22:
// one correctly rounded division / multiplication) and strtod otherwise.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fc_common.h"

namespace fc {

static const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// appends x formatted like "%{width}.{dec}f" (dec = 6 or 8), optional ' ' flag
static void append_fixed(std::string &out, double x, int width, int dec, bool space_flag) {
  char buf[64];
  int n = -1;
  const double ax = std::fabs(x);
  if (std::isfinite(x) && ax < 4.0e7) {  // ax * 10^8 < 2^52: every integer nearby is exact
    const double scale = kPow10[dec];
    const double p = ax * scale;
    const double e = std::fma(ax, scale, -p);  // ax*scale == p + e exactly
    double r = std::nearbyint(p);              // current mode: round-half-even
    const double d = (p - r) + e;              // exact offset of the true product from r (|d| <~ 0.5)
    bool sure = true;
    if (std::fabs(std::fabs(d) - 0.5) < 1e-9) sure = false;  // too close to a tie to call here
    else if (d > 0.5) r += 1.0;
    else if (d < -0.5) r -= 1.0;
    if (sure) {
      unsigned long long v = (unsigned long long)r;
      const unsigned long long sc = (unsigned long long)scale;
      unsigned long long ip = v / sc, fp = v % sc;
      char tmp[48];
      int k = 0;
      for (int i = 0; i < dec; ++i) {
        tmp[k++] = (char)('0' + fp % 10);
        fp /= 10;
      }
      tmp[k++] = '.';
      do {
        tmp[k++] = (char)('0' + ip % 10);
        ip /= 10;
      } while (ip);
      if (std::signbit(x)) tmp[k++] = '-';
      else if (space_flag) tmp[k++] = ' ';
      n = 0;
      for (int pad = width - k; pad > 0; --pad) buf[n++] = ' ';
      while (k) buf[n++] = tmp[--k];
    }
  }
  if (n < 0) n = std::snprintf(buf, sizeof buf, space_flag ? "% *.*f" : "%*.*f", width, dec, x);
  out.append(buf, (size_t)n);
}

static void append_int(std::string &out, long long v) {
  char buf[32];
  const int n = std::snprintf(buf, sizeof buf, "%lld", v);
  out.append(buf, (size_t)n);
}

int xyz_write(const char *path, const char *const *atoms, int64_t A, const double *coords, int64_t N,
              const char *label, int mode) {
  FILE *f = std::fopen(path, "wb");
  if (!f) return set_error(FC_E_INVALID, "cannot open %s for writing: %s", path, std::strerror(errno));
  std::string out;
  out.reserve(1 << 22);
  for (int64_t n = 0; n < N; ++n) {
    const double *x = coords + n * A * 3;
    if (mode == 0) {
      if (n) out.push_back('\n');
      append_int(out, A);
      out.append("\nExported from FIRECODE Ensemble (");
      out.append(label);
      out.append(")\n");
      for (int64_t a = 0; a < A; ++a) {
        if (a) out.push_back('\n');
        out.append(atoms[a]);
        for (int c = 0; c < 3; ++c) {
          out.push_back(' ');
          append_fixed(out, x[a * 3 + c], 15, 8, false);
        }
      }
    } else {
      append_int(out, A);
      out.push_back('\n');
      out.append(label);
      out.push_back('\n');
      for (int64_t a = 0; a < A; ++a) {
        out.append(atoms[a]);
        out.append("    ");
        for (int c = 0; c < 3; ++c) {
          out.push_back(' ');
          append_fixed(out, x[a * 3 + c], 0, 6, true);
        }
        out.push_back('\n');
      }
    }
    if (out.size() > (1u << 22) - 4096) {
      if (std::fwrite(out.data(), 1, out.size(), f) != out.size()) {
        std::fclose(f);
        return set_error(FC_E_INVALID, "short write to %s", path);
      }
      out.clear();
    }
  }
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  if (std::fclose(f) != 0 || !ok) return set_error(FC_E_INVALID, "short write to %s", path);
  return FC_OK;
}

// ---- reader ---------------------------------------------------------------------
static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

// one token -> double, exactly as Python's float(token) for plain decimals
static bool parse_double(const char *s, const char *end, double &out) {
  const char *p = s;
  bool neg = false;
  if (p < end && (*p == '+' || *p == '-')) neg = (*p++ == '-');
  unsigned long long m = 0;
  int digits = 0, frac = 0;
  bool seen_dot = false, any = false, simple = true;
  for (; p < end; ++p) {
    const char c = *p;
    if (c >= '0' && c <= '9') {
      any = true;
      if (digits < 19) {
        m = m * 10 + (unsigned)(c - '0');
        if (m || digits) ++digits;
        if (seen_dot) ++frac;
      } else {
        simple = false;  // too many digits for the exact path
      }
    } else if (c == '.' && !seen_dot) {
      seen_dot = true;
    } else {
      simple = false;  // exponent, nan, inf, underscores ...: leave it to strtod
      break;
    }
  }
  if (simple && any && m < (1ull << 53) && frac <= 22) {
    double v = (double)m;
    if (frac) v /= kPow10[frac];
    out = neg ? -v : v;
    return true;
  }
  std::string tok(s, end);
  char *e = nullptr;
  errno = 0;
  const double v = std::strtod(tok.c_str(), &e);
  if (e == tok.c_str() || *e != '\0') return false;
  out = v;
  return true;
}

struct LineReader {
  const char *p, *end;
  bool next(const char *&b, const char *&e) {  // one line without its '\n'
    if (p >= end) return false;
    b = p;
    const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
    e = nl ? nl : end;
    p = nl ? nl + 1 : end;
    return true;
  }
};

static bool blank(const char *b, const char *e) {
  for (; b < e; ++b)
    if (!is_space(*b)) return false;
  return true;
}

static int slurp(const char *path, std::vector<char> &buf) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return set_error(FC_E_INVALID, "cannot open %s: %s", path, std::strerror(errno));
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf.resize((size_t)(sz > 0 ? sz : 0));
  const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
  std::fclose(f);
  if (got != buf.size()) return set_error(FC_E_INVALID, "short read from %s", path);
  return FC_OK;
}

// pass 1 (coords_out == nullptr): count conformers, atoms of the first one.
// pass 2: fill atoms_out (A x 8 chars, NUL padded, first conformer) and coords_out (N, A, 3).
int xyz_read(const char *path, int64_t *N_io, int64_t *A_io, char *atoms_out, double *coords_out) {
  std::vector<char> buf;
  FC_TRY(slurp(path, buf));
  LineReader lr{buf.data(), buf.data() + buf.size()};
  const char *b, *e;
  int64_t n = 0, A0 = -1;
  const bool fill = coords_out != nullptr;
  while (lr.next(b, e)) {
    if (blank(b, e)) continue;
    // int(num): surrounding whitespace allowed
    std::string tok(b, e);
    char *endp = nullptr;
    const long long cnt = std::strtoll(tok.c_str(), &endp, 10);
    while (endp && *endp && is_space(*endp)) ++endp;
    if (endp == tok.c_str() || (endp && *endp != '\0') || cnt < 0)
      return set_error(FC_E_INVALID, "%s: expected an atom count, got '%s'", path, tok.c_str());
    if (!lr.next(b, e)) break;  // comment line; truncated file: the reference stops silently
    if (A0 < 0) A0 = cnt;
    if (fill && n >= *N_io) break;  // what follows is the truncated conformer the scan dropped
    if (fill && cnt != *A_io)
      return set_error(FC_E_INVALID, "%s: conformer %lld has %lld atoms, expected %lld", path,
                       (long long)n, cnt, (long long)*A_io);
    bool complete = true;
    for (long long a = 0; a < cnt; ++a) {
      if (!lr.next(b, e)) {
        complete = false;
        break;
      }
      if (!fill && n > 0) continue;  // counting pass: only the first conformer is inspected
      // tokens: symbol x y z [...]
      const char *q = b;
      const char *tb[4], *te[4];
      int nt = 0;
      while (q < e && nt < 4) {
        while (q < e && is_space(*q)) ++q;
        if (q >= e) break;
        tb[nt] = q;
        while (q < e && !is_space(*q)) ++q;
        te[nt++] = q;
      }
      if (nt < 4) return set_error(FC_E_INVALID, "%s: malformed atom line in conformer %lld", path, (long long)n);
      if (fill) {
        if (n == 0) {
          const size_t len = (size_t)(te[0] - tb[0]);
          std::memset(atoms_out + a * 8, 0, 8);
          std::memcpy(atoms_out + a * 8, tb[0], len < 7 ? len : 7);
        }
        for (int c = 0; c < 3; ++c) {
          double v;
          if (!parse_double(tb[c + 1], te[c + 1], v))
            return set_error(FC_E_INVALID, "%s: cannot parse a coordinate in conformer %lld", path, (long long)n);
          coords_out[(n * *A_io + a) * 3 + c] = v;
        }
      }
    }
    if (!complete) break;  // StopIteration inside a conformer: it is dropped (ensemble.py:91-92)
    ++n;
  }
  if (!fill) {
    *N_io = n;
    *A_io = A0 < 0 ? 0 : A0;
  } else if (n != *N_io) {
    return set_error(FC_E_INVALID, "%s: found %lld conformers, expected %lld", path, (long long)n, (long long)*N_io);
  }
  return FC_OK;
}

}  // namespace fc
