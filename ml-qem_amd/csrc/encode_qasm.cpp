// Host-side (CPU) native encoder: OpenQASM-2 text -> the op-node feature matrix and the op->op qubit-wire edge list
// of the reference's circuit_to_graph_data_json (blackwater/data/utils.py:198-389), i.e. exactly what
// ExpValueEntry.to_pyg_data consumes (blackwater/data/generators/exp_val.py:63-70).  Same node order (program order),
// same feature layout, same edge order (for each source op, its out-edges most recently inserted first), same doubles.
// No DAG library: a DAG built by appending ops has one chain per wire, so the edge list follows from "the previous op
// on each wire".  Pure C++17, no HIP calls: it can run on a box without a GPU.
//
// The decorators call this once per circuit of every run() (ngem/estimator.py:49-84), on 100-qubit circuits of ~2e4
// statements, so the scan allocates nothing per statement: statements, names and arguments are views into the text, an op
// is seven integers, qubit lists and parameters live in two pools, gate names are interned, and the per-source out-lists
// are a linked list threaded through one array (2.7 us -> 0.2 us per statement against a std::string-per-token version).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include <pthread.h>

#include "../../include/mlqem_hip.h"

namespace {

using View = std::string_view;

struct Op {
  int type;                 // index into Circuit::names
  int q_off, q_cnt;         // flat qubit indices: qpool[q_off .. q_off + q_cnt)
  int c_off, c_cnt;         // flat clbit indices, same pool
  int p_off, p_cnt;         // parameters: ppool[p_off .. p_off + p_cnt)
};

struct Reg { int base, size; };

// Two kinds of failure: text that is not OpenQASM 2 (MLQEM_ERR_BAD_ARG) and a well-formed circuit the encoding does not
// cover (a gate outside gates_set, more than 3 qubits / parameters: MLQEM_ERR_UNSUPPORTED, where the reference raises too).
struct ParseError { std::string what; bool unsupported = false; };

// Limits that keep hostile text from turning into unbounded recursion or allocation: the text arrives through a public
// entry point (the decorators hand over whatever the caller's circuits serialise to).
constexpr int kMaxExprDepth = 64;           // nested parentheses / function calls / unary signs in one angle expression
constexpr long kMaxRegisterBits = 1 << 20;  // qubits + clbits of a circuit (the largest devices have ~1e3)

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_alpha(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
inline bool is_word(char c) { return is_alpha(c) || is_digit(c) || c == '_'; }

inline View trim(View s) {
  size_t a = 0, b = s.size();
  while (a < b && is_space(s[a])) ++a;
  while (b > a && is_space(s[b - 1])) --b;
  return s.substr(a, b - a);
}

inline bool starts_with(View s, const char* pre) { const size_t n = std::strlen(pre); return s.size() >= n && std::memcmp(s.data(), pre, n) == 0; }

std::string str(View v) { return std::string(v.data(), v.size()); }
std::string head64(View v) { return str(v.substr(0, 64)); }

// decimal integer followed by ']' at the start of s (after optional spaces); -1 when it is anything else or does not fit
long bracket_index(View s) {
  size_t i = 0;
  while (i < s.size() && is_space(s[i])) ++i;
  if (i >= s.size() || !is_digit(s[i])) return -1;
  long v = 0;
  for (; i < s.size() && is_digit(s[i]); ++i) {
    v = v * 10 + (s[i] - '0');
    if (v > kMaxRegisterBits) return -1;
  }
  while (i < s.size() && is_space(s[i])) ++i;
  return (i < s.size() && s[i] == ']') ? v : -1;
}

// ---- angle expressions: numbers, pi, + - * / ^, unary sign, parentheses, sin cos tan exp ln sqrt asin acos atan.
// Works on [p, end) of the statement text; the byte at `end` is a delimiter of the enclosing statement (',' ')' ';' or the
// terminating NUL), so strtod cannot run past it.
struct Expr {
  const char* p;
  const char* end;
  int depth = 0;
  struct Nest {   // one level of recursion (a parenthesis, a function argument, an exponent)
    int& d;
    explicit Nest(int& depth) : d(depth) { if (++d > kMaxExprDepth) throw ParseError{"angle expression nested too deeply"}; }
    ~Nest() { --d; }
  };
  explicit Expr(View s) : p(s.data()), end(s.data() + s.size()) {}
  char cur() const { return p < end ? *p : '\0'; }
  void ws() { while (p < end && is_space(*p)) ++p; }
  double parse() { double v = sum(); ws(); if (p < end) throw ParseError{"trailing characters in angle expression"}; return v; }
  double sum() {
    double v = product();
    for (;;) {
      ws();
      if (cur() == '+') { ++p; v = v + product(); }
      else if (cur() == '-') { ++p; v = v - product(); }
      else return v;
    }
  }
  double product() {
    double v = unary();
    for (;;) {
      ws();
      if (cur() == '*') { ++p; v = v * unary(); }
      else if (cur() == '/') { ++p; v = v / unary(); }
      else return v;
    }
  }
  double unary() {
    bool neg = false;                       // a run of signs is a loop, not a recursion
    for (ws(); cur() == '-' || cur() == '+'; ws()) { if (*p == '-') neg = !neg; ++p; }
    const double v = power();
    return neg ? -v : v;
  }
  double power() {
    double b = atom();
    ws();
    if (cur() == '^') { ++p; Nest n(depth); return std::pow(b, unary()); }
    return b;
  }
  double atom() {
    ws();
    const char c = cur();
    if (c == '(') { Nest n(depth); ++p; double v = sum(); ws(); if (cur() != ')') throw ParseError{"missing ) in angle expression"}; ++p; return v; }
    if (is_digit(c) || c == '.') {
      char* stop;
      double v = std::strtod(p, &stop);
      if (stop == p || stop > end) throw ParseError{"bad number in angle expression"};
      p = stop;
      return v;
    }
    if (is_alpha(c)) {
      const char* b = p;
      while (p < end && is_word(*p)) ++p;
      const View id(b, (size_t)(p - b));
      if (id == "pi") return M_PI;
      ws();
      if (cur() != '(') throw ParseError{"unknown identifier '" + head64(id) + "' in angle expression"};
      Nest n(depth);
      ++p; double a = sum(); ws(); if (cur() != ')') throw ParseError{"missing ) after function"}; ++p;
      if (id == "sin") return std::sin(a); if (id == "cos") return std::cos(a); if (id == "tan") return std::tan(a);
      if (id == "exp") return std::exp(a); if (id == "ln") return std::log(a); if (id == "sqrt") return std::sqrt(a);
      if (id == "asin") return std::asin(a); if (id == "acos") return std::acos(a); if (id == "atan") return std::atan(a);
      throw ParseError{"unknown function '" + head64(id) + "'"};
    }
    throw ParseError{"bad angle expression"};
  }
};

// the pieces of s between commas outside parentheses, trimmed; "a," has an (empty) second element: the callers reject it
void split_top(View s, std::vector<View>& out) {
  out.clear();
  int depth = 0;
  size_t from = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (c == '(') ++depth; else if (c == ')') --depth;
    else if (c == ',' && depth == 0) { out.push_back(trim(s.substr(from, i - from))); from = i + 1; }
  }
  const View last = trim(s.substr(from));
  if (!last.empty() || !out.empty()) out.push_back(last);
}

struct Circuit {
  int nq = 0, nc = 0;
  std::vector<int> reg_index;            // register-local index of every flat qubit (qiskit's Qubit.index)
  std::vector<Op> ops;
  std::vector<int> bits;                 // qubit / clbit pool
  std::vector<double> params;            // parameter pool
  std::deque<std::string> names;         // interned op names (deque: the views in name_id stay valid)
  std::unordered_map<View, int> name_id;
  int barrier = -1, measure = -1, reset = -1;

  int intern(View name) {
    auto it = name_id.find(name);
    if (it != name_id.end()) return it->second;
    names.emplace_back(name.data(), name.size());
    const int id = (int)names.size() - 1;
    name_id.emplace(View(names.back()), id);
    return id;
  }
  const int* qubits(const Op& o) const { return bits.data() + o.q_off; }
  const int* clbits(const Op& o) const { return bits.data() + o.c_off; }
  void clear() {     // keeps the vectors' capacity: a worker thread parses circuit after circuit into one scratch Circuit
    nq = nc = 0; reg_index.clear(); ops.clear(); bits.clear(); params.clear(); name_id.clear(); names.clear();
    barrier = measure = reset = -1;
  }
  // An exactly-sized copy to keep (without the interning table, whose keys are views into THIS circuit's names): one
  // allocation per array instead of the chain of doublings a vector grown statement by statement goes through.
  Circuit compact() const {
    Circuit c;
    c.nq = nq; c.nc = nc; c.reg_index = reg_index; c.ops = ops; c.bits = bits; c.params = params; c.names = names;
    c.barrier = barrier; c.measure = measure; c.reset = reset;
    return c;
  }
};

struct Registers {
  std::unordered_map<std::string, Reg> map;
  std::string last_name;                 // nearly every argument of a circuit names the same register
  Reg last{0, 0};
  bool has_last = false;
  const Reg* find(View name) {
    if (has_last && name == last_name) return &last;
    auto it = map.find(str(name));
    if (it == map.end()) return nullptr;
    last_name = it->first; last = it->second; has_last = true;
    return &last;
  }
};

// appends the flat indices `arg` names (one bit, or a whole register) to out; returns how many
int bits_of(View arg, Registers& regs, std::vector<int>& out) {
  const View a = trim(arg);
  const size_t br = a.find('[');
  const View name = trim(br == View::npos ? a : a.substr(0, br));
  const Reg* r = regs.find(name);
  if (!r) throw ParseError{"unknown register in '" + head64(a) + "'"};
  if (br == View::npos) { for (int i = 0; i < r->size; ++i) out.push_back(r->base + i); return r->size; }
  const long idx = bracket_index(a.substr(br + 1));
  if (idx < 0 || idx >= r->size) throw ParseError{"bad or out-of-range index in '" + head64(a) + "'"};
  out.push_back(r->base + (int)idx);
  return 1;
}

// ---- the common statement, fast: `name q[i];`, `name(1.5707963267948966) q[i];`, `name q[i],q[j];` -------------------------
// A transpiled circuit is tens of thousands of these (20 k statements per 100-qubit Trotter circuit, a run() of a VQE loop hands
// over a thousand circuits: blackwater/library/ngem/estimator.py:49-84), and the general path spends its time in machinery they
// do not need: two vectors of argument views, a hash of the gate name, strtod on a 17-digit literal.  Anything else -- angle
// expressions, whole-register arguments, unknown registers, out-of-range indices, trailing text -- returns false and goes through
// the general path, which also words the error messages.

// A decimal literal [digits][.digits][e[+-]digits] at p (no sign) -> its correctly rounded double, or nullptr (then the caller
// uses strtod).  Up to 19 significant digits are an exact 64-bit integer m; m * 10^e is exact in one operation when m < 2^53 and
// |e| <= 22 (both operands exact doubles: one correctly rounded IEEE operation).  Otherwise, on x86-64, the operation is done in
// the 80-bit format (m and 10^|e| <= 10^27 are exact there) and rounded to double a second time -- which is the correct rounding
// unless the 64-bit mantissa sits within two units of a double's rounding boundary (the low 11 bits near 0x400): those literals
// take strtod.  Checked against strtod on random and adversarial literals (fuzz/fuzz_encode_qasm.cpp, tests/test_native_encoder.py).
const char* fast_literal(const char* p, const char* end, double& out) {
  uint64_t m = 0;
  int sig = 0, frac = 0, digits = 0;
  bool dropped = false;
  const char* q = p;
  for (; q < end && is_digit(*q); ++q, ++digits) {
    if (sig < 19) { m = m * 10 + (uint64_t)(*q - '0'); if (m) ++sig; } else dropped = true;
  }
  if (q < end && *q == '.') {
    ++q;
    for (; q < end && is_digit(*q); ++q, ++digits) {
      if (sig < 19) { m = m * 10 + (uint64_t)(*q - '0'); if (m) ++sig; ++frac; } else dropped = true;
    }
  }
  if (digits == 0 || dropped) return nullptr;
  int e10 = 0;
  if (q < end && (*q == 'e' || *q == 'E')) {
    const char* r = q + 1;
    bool neg = false;
    if (r < end && (*r == '+' || *r == '-')) { neg = *r == '-'; ++r; }
    if (r >= end || !is_digit(*r)) return nullptr;
    for (; r < end && is_digit(*r); ++r) { e10 = e10 * 10 + (*r - '0'); if (e10 > 400) return nullptr; }
    if (neg) e10 = -e10;
    q = r;
  }
  e10 -= frac;
  if (m == 0) { out = 0.0; return q; }
  static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20,
                                 1e21, 1e22};
  if (m < (1ull << 53) && e10 >= -22 && e10 <= 22) {
    out = e10 < 0 ? (double)m / p10[-e10] : (double)m * p10[e10];
    return q;
  }
#if defined(__x86_64__) && defined(__LDBL_MANT_DIG__) && __LDBL_MANT_DIG__ == 64
  if (e10 >= -27 && e10 <= 27) {
    static const long double p10l[28] = {1e0L, 1e1L, 1e2L, 1e3L, 1e4L, 1e5L, 1e6L, 1e7L, 1e8L, 1e9L, 1e10L, 1e11L, 1e12L, 1e13L, 1e14L, 1e15L, 1e16L, 1e17L,
                                         1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
    const long double r = e10 < 0 ? (long double)m / p10l[-e10] : (long double)m * p10l[e10];
    uint64_t mant;
    std::memcpy(&mant, &r, sizeof mant);                  // the explicit 64-bit significand of the x87 format
    const unsigned low = (unsigned)(mant & 0x7FFu);
    if (low >= 0x3FEu && low <= 0x402u) return nullptr;   // too close to a double's rounding boundary to round twice
    out = (double)r;                                      // 19 digits x 10^+-27: far inside the normal range of double
    return q;
  }
#endif
  return nullptr;
}

// the last few gate names of a circuit, compared by length and bytes: no hash per statement
struct NameCache {
  static constexpr int kSlots = 8;
  const char* ptr[kSlots]; size_t len[kSlots]; int id[kSlots];
  int used = 0, next = 0;
  void reset() { used = next = 0; }
};

int intern_cached(Circuit& c, NameCache& cache, View name) {
  for (int k = 0; k < cache.used; ++k)
    if (cache.len[k] == name.size() && std::memcmp(cache.ptr[k], name.data(), name.size()) == 0) return cache.id[k];
  const int id = c.intern(name);
  const std::string& kept = c.names[(size_t)id];        // a deque element: its bytes stay where they are
  const int slot = cache.used < NameCache::kSlots ? cache.used++ : (cache.next++ % NameCache::kSlots);
  cache.ptr[slot] = kept.data(); cache.len[slot] = kept.size(); cache.id[slot] = id;
  return id;
}

// what the statement loop below recognises by prefix (and `gate` / `opaque` / `if`, which are no applications of a gate)
inline bool is_keyword(View w) {
  switch (w[0]) {
    case 'O': return starts_with(w, "OPENQASM");
    case 'i': return starts_with(w, "include") || w == "if";
    case 'q': return starts_with(w, "qreg");
    case 'c': return starts_with(w, "creg");
    case 'm': return starts_with(w, "measure");
    case 'b': return starts_with(w, "barrier");
    case 'r': return starts_with(w, "reset");
    case 'g': return w == "gate";
    case 'o': return w == "opaque";
    default: return false;
  }
}

// `st` (trimmed, non-empty) as one gate on explicit bits; false = not of that form (nothing was appended to c)
bool fast_gate(View st, Circuit& c, Registers& qregs, NameCache& cache) {
  const char* p = st.data();
  const char* const end = p + st.size();
  const char* w = p;
  while (w < end && is_word(*w)) ++w;
  if (w == p || is_digit(*p)) return false;
  const View name(p, (size_t)(w - p));
  if (is_keyword(name)) return false;
  double param = 0.0;
  int p_cnt = 0;
  p = w;
  while (p < end && is_space(*p)) ++p;
  if (p < end && *p == '(') {
    ++p;
    while (p < end && is_space(*p)) ++p;
    bool neg = false;
    if (p < end && *p == '-') { neg = true; ++p; while (p < end && is_space(*p)) ++p; }
    const char* after = fast_literal(p, end, param);
    if (!after) return false;
    p = after;
    while (p < end && is_space(*p)) ++p;
    if (p >= end || *p != ')') return false;              // an expression, a second parameter: the general path
    ++p;
    if (neg) param = -param;
    p_cnt = 1;
    while (p < end && is_space(*p)) ++p;
  }
  int bits[3], n_bits = 0;
  for (;;) {
    const char* r = p;
    while (r < end && is_word(*r)) ++r;
    if (r == p || n_bits == 3) return false;
    const View reg(p, (size_t)(r - p));
    p = r;
    while (p < end && is_space(*p)) ++p;
    if (p >= end || *p != '[') return false;              // a whole-register argument broadcasts: the general path
    ++p;
    while (p < end && is_space(*p)) ++p;
    if (p >= end || !is_digit(*p)) return false;
    long idx = 0;
    for (; p < end && is_digit(*p); ++p) { idx = idx * 10 + (*p - '0'); if (idx > kMaxRegisterBits) return false; }
    while (p < end && is_space(*p)) ++p;
    if (p >= end || *p != ']') return false;
    ++p;
    const Reg* rg = qregs.find(reg);
    if (!rg || idx >= rg->size) return false;             // the general path words the error
    bits[n_bits++] = rg->base + (int)idx;
    while (p < end && is_space(*p)) ++p;
    if (p == end) break;
    if (*p != ',') return false;
    ++p;
    while (p < end && is_space(*p)) ++p;
  }
  const int type = intern_cached(c, cache, name);
  const int q_off = (int)c.bits.size();
  for (int k = 0; k < n_bits; ++k) c.bits.push_back(bits[k]);
  const int p_off = (int)c.params.size();
  if (p_cnt) c.params.push_back(param);
  c.ops.push_back(Op{type, q_off, n_bits, q_off, 0, p_off, p_cnt});
  return true;
}

// ---- the same statements scanned a machine word at a time (round 5) ------------------------------------------------------------
// fast_gate above still walks views, calls memchr / memcmp per token and parses every literal: 40 ns per statement of ~20 bytes,
// 11 M statements per run() of 1024 100-qubit circuits.  The scanner below reads the text where it lies, from the first byte of a
// statement to its ';':
//   * names (gate, register) of up to 8 bytes are ONE unaligned 64-bit load masked to their length: a keyword test is a masked
//     compare, the gate-name and register caches compare integers, not bytes;
//   * the parenthesised parameter is looked up by its bytes -- length, first and last 8 bytes, the 8 in the middle -- in a small
//     table before anything is parsed: a transpiled circuit repeats a handful of angles thousands of times (six distinct literals
//     among 9 930 in a 10-step Trotter circuit); a miss parses with fast_literal as before and remembers the value;
//   * nothing is appended to the circuit before the ';' is reached, so any other shape (expressions, whole registers, unknown
//     names, out-of-range indices, names longer than 8 bytes, the last 8 bytes of the text) returns false and the statement goes
//     through fast_gate / the general path unchanged, which also word the errors.
// Same circuits as the general path, op for op (tests/test_native_encoder.py runs all three against each other; the fuzz driver
// feeds malformed text through this entry first).
inline uint64_t load8(const char* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline uint64_t low_bytes(int n) { return n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1); }
constexpr uint64_t word8(const char* s) {      // the bytes of a short literal as load8 sees them (little endian)
  uint64_t v = 0;
  for (int i = 0; i < 8 && s[i]; ++i) v |= (uint64_t)(unsigned char)s[i] << (8 * i);
  return v;
}
// is_keyword() on the word: prefixes of OPENQASM / include / qreg / creg / measure / barrier / reset, exactly if / gate / opaque
inline bool keyword_word(uint64_t w, int n) {
  switch ((char)(w & 0xFF)) {
    case 'O': return w == word8("OPENQASM");
    case 'i': return (w & low_bytes(7)) == word8("include") || (n == 2 && w == word8("if"));
    case 'q': return (w & low_bytes(4)) == word8("qreg");
    case 'c': return (w & low_bytes(4)) == word8("creg");
    case 'm': return (w & low_bytes(7)) == word8("measure");
    case 'b': return (w & low_bytes(7)) == word8("barrier");
    case 'r': return (w & low_bytes(5)) == word8("reset");
    case 'g': return n == 4 && w == word8("gate");
    case 'o': return n == 6 && w == word8("opaque");
    default: return false;
  }
}

struct WordCache {      // per circuit: gate names and the register last named, as words; parameters by their bytes
  static constexpr int kNames = 8, kParams = 64;
  uint64_t name[kNames]; int name_id[kNames]; int names = 0, next_name = 0;
  uint64_t reg_word = 0; int reg_len = -1; Reg reg{0, 0};
  struct Param { uint64_t head, mid, tail; int len; double value; };
  Param param[kParams];
  void reset() { names = next_name = 0; reg_len = -1; for (auto& q : param) q.len = -1; }
};

// a word of at most 8 word-characters at p (p + 8 <= end): its length, 0 when it does not start one or is longer
inline int word_at(const char* p, uint64_t& w) {
  if (!(is_alpha(*p) || *p == '_')) return 0;
  int n = 1;
  while (n < 8 && is_word(p[n])) ++n;
  if (n == 8 && is_word(p[8])) return 0;      // longer than a machine word (p[8] is readable: the caller keeps 9 bytes)
  w = load8(p) & low_bytes(n);
  return n;
}

// One statement starting at p (first non-space byte) -> appended to c, p behind its ';'.  false: nothing appended, p unchanged.
bool fast_statement(const char*& pos, const char* const end, Circuit& c, Registers& qregs, WordCache& wc) {
  const char* p = pos;
  if (end - p < 16) return false;                       // the loads below read up to 9 bytes ahead of a token's start
  uint64_t w;
  const int n = word_at(p, w);
  if (n == 0 || keyword_word(w, n)) return false;
  const char* const name_at = p;
  p += n;
  while (p < end && is_space(*p)) ++p;
  double param = 0.0;
  int p_cnt = 0;
  if (p < end && *p == '(') {
    const char* const lp = p + 1;
    // the closing parenthesis, eight bytes at a time, within 64 bytes (a hit below means the bytes in between ARE a literal seen
    // before; a miss parses them, and anything but a literal fails there)
    const char* rp = nullptr;
    for (const char* q = lp; q + 8 <= end && q < lp + 64; q += 8) {
      const uint64_t x = load8(q) ^ 0x2929292929292929ull;              // ')' = 0x29: a zero byte where one is
      const uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
      if (z) { rp = q + (__builtin_ctzll(z) >> 3); break; }
    }
    if (!rp) return false;                              // unbalanced, or far away: the general path
    const int len = (int)(rp - lp);
    if (len < 1) return false;
    // the bytes between the parentheses as (length, first 8, middle 8, last 8): exact for up to 24 bytes
    uint64_t head = 0, mid = 0, tail = 0;
    const bool keyed = len <= 24 && end - lp >= 8;
    if (keyed) {
      if (len >= 8) { head = load8(lp); tail = load8(rp - 8); if (len > 16) mid = load8(lp + 8); }
      else { head = load8(lp) & low_bytes(len); tail = head; }
    }
    WordCache::Param* slot = nullptr;
    if (keyed) {
      slot = &wc.param[((head * 0x9E3779B97F4A7C15ull) ^ (tail * 0xC2B2AE3D27D4EB4Full) ^ (uint64_t)len) >> 58];
      if (slot->len == len && slot->head == head && slot->tail == tail && slot->mid == mid) { param = slot->value; p_cnt = 1; }
    }
    if (!p_cnt) {
      const char* q = lp;
      while (q < rp && is_space(*q)) ++q;
      bool neg = false;
      if (q < rp && *q == '-') { neg = true; ++q; while (q < rp && is_space(*q)) ++q; }
      const char* after = fast_literal(q, rp, param);
      if (!after) return false;
      while (after < rp && is_space(*after)) ++after;
      if (after != rp) return false;                    // an expression, a second parameter: the general path
      if (neg) param = -param;
      p_cnt = 1;
      if (slot) *slot = WordCache::Param{head, mid, tail, len, param};
    }
    p = rp + 1;
    while (p < end && is_space(*p)) ++p;
  }
  int bits[3], n_bits = 0;
  for (;;) {
    if (end - p < 16 || n_bits == 3) return false;
    uint64_t rw;
    const int rn = word_at(p, rw);
    if (rn == 0) return false;
    const Reg* rg;
    if (rn == wc.reg_len && rw == wc.reg_word) rg = &wc.reg;
    else {
      rg = qregs.find(View(p, (size_t)rn));
      if (!rg) return false;
      wc.reg = *rg; wc.reg_word = rw; wc.reg_len = rn;
      rg = &wc.reg;
    }
    p += rn;
    while (p < end && is_space(*p)) ++p;
    if (p >= end || *p != '[') return false;            // a whole-register argument broadcasts: the general path
    ++p;
    while (p < end && is_space(*p)) ++p;
    if (p >= end || !is_digit(*p)) return false;
    long idx = 0;
    for (; p < end && is_digit(*p); ++p) { idx = idx * 10 + (*p - '0'); if (idx > kMaxRegisterBits) return false; }
    while (p < end && is_space(*p)) ++p;
    if (p >= end || *p != ']') return false;
    ++p;
    if (idx >= rg->size) return false;                  // the general path words the error
    bits[n_bits++] = rg->base + (int)idx;
    while (p < end && is_space(*p)) ++p;
    if (p >= end) return false;                         // no ';': the last statement of a text goes the general way
    if (*p == ';') break;
    if (*p != ',') return false;
    ++p;
    while (p < end && is_space(*p)) ++p;
  }
  // the gate name -> its id (after the arguments: an unknown register must not intern a name)
  int type = -1;
  for (int k = 0; k < wc.names; ++k)
    if (wc.name[k] == w) { type = wc.name_id[k]; break; }
  if (type < 0) {
    type = c.intern(View(name_at, (size_t)n));
    const int slot = wc.names < WordCache::kNames ? wc.names++ : (wc.next_name++ % WordCache::kNames);
    wc.name[slot] = w; wc.name_id[slot] = type;
  }
  const int q_off = (int)c.bits.size();
  for (int k = 0; k < n_bits; ++k) c.bits.push_back(bits[k]);
  const int p_off = (int)c.params.size();
  if (p_cnt) c.params.push_back(param);
  c.ops.push_back(Op{type, q_off, n_bits, q_off, 0, p_off, p_cnt});
  pos = p + 1;
  return true;
}

// Parses `text` into c (cleared first; its capacity is reused).  `buf_a` / `buf_b` are scratch strings for the stripped text.
void parse_qasm(const char* text, Circuit& c, std::string& s, std::string& t) {
  const size_t len = std::strlen(text);
  View all(text, len);                   // text without comments and definitions is scanned in place
  if (std::strstr(text, "//")) {         // strip // comments
    s.clear();
    s.reserve(len + 1);
    for (size_t i = 0; i < len;) {
      if (text[i] == '/' && i + 1 < len && text[i + 1] == '/') { while (i < len && text[i] != '\n') ++i; continue; }
      s.push_back(text[i++]);
    }
    all = View(s);
  }
  // drop gate / opaque definitions (kept opaque: an op using one carries the definition's name)
  if (all.find("gate ") != View::npos || all.find("opaque ") != View::npos) {
    t.clear();
    t.reserve(all.size() + 1);
    const char* src = all.data();
    for (size_t i = 0; i < all.size();) {
      const char ch = src[i];
      if ((ch == 'g' || ch == 'o') && (i == 0 || !is_word(src[i - 1])) &&
          (all.compare(i, 5, "gate ") == 0 || all.compare(i, 7, "opaque ") == 0)) {
        const size_t brace = all.find('{', i), semi = all.find(';', i);
        if (brace != View::npos && (semi == View::npos || brace < semi)) {
          const size_t close = all.find('}', brace);
          if (close == View::npos) throw ParseError{"unterminated gate definition"};
          i = close + 1;
        } else {
          if (semi == View::npos) throw ParseError{"unterminated opaque declaration"};
          i = semi + 1;
        }
        continue;
      }
      t.push_back(ch);
      ++i;
    }
    all = View(t);
  }
  c.clear();
  c.barrier = c.intern("barrier"); c.measure = c.intern("measure"); c.reset = c.intern("reset");
  Registers qregs, cregs;
  NameCache name_cache;
  name_cache.reset();
  // MLQEM_QASM_FAST: 0 = the general path only, 1 = fast_gate (round 4), default 2 = the word scanner in front of it
  static const int fast_level = std::getenv("MLQEM_QASM_FAST") ? std::atoi(std::getenv("MLQEM_QASM_FAST")) : 2;
  static const bool fast_path = fast_level != 0;
  WordCache words;
  words.reset();
  std::vector<View> pieces;
  std::vector<int> arg_off, arg_cnt, scratch;
  size_t pos = 0;
  const char* const text_end = all.data() + all.size();
  while (pos < all.size()) {
    if (fast_level >= 2) {
      const char* q = all.data() + pos;
      while (q < text_end && is_space(*q)) ++q;
      const char* const from = q;
      while (fast_statement(q, text_end, c, qregs, words))
        while (q < text_end && is_space(*q)) ++q;
      if (q != from) { pos = (size_t)(q - all.data()); if (pos >= all.size()) break; }
    }
    const size_t semi = all.find(';', pos);
    const View st = trim(all.substr(pos, semi == View::npos ? View::npos : semi - pos));
    pos = semi == View::npos ? all.size() : semi + 1;
    if (st.empty()) continue;
    if (fast_path && fast_gate(st, c, qregs, name_cache)) continue;
    if (starts_with(st, "OPENQASM") || starts_with(st, "include")) continue;
    if (starts_with(st, "qreg") || starts_with(st, "creg")) {
      const bool q = st[0] == 'q';
      const View rest = trim(st.substr(4));
      const size_t br = rest.find('[');
      if (br == View::npos) throw ParseError{"bad register declaration '" + head64(st) + "'"};
      const View name = trim(rest.substr(0, br));
      const long size_l = bracket_index(rest.substr(br + 1));
      if (name.empty() || size_l < 0 || c.nq + c.nc + size_l > kMaxRegisterBits) throw ParseError{"bad register declaration '" + head64(st) + "'"};
      const std::string key = str(name);
      if (qregs.map.count(key) || cregs.map.count(key)) throw ParseError{"register '" + head64(name) + "' declared twice"};
      const int size = (int)size_l;
      if (q) { qregs.map[key] = Reg{c.nq, size}; c.nq += size; for (int i = 0; i < size; ++i) c.reg_index.push_back(i); }
      else { cregs.map[key] = Reg{c.nc, size}; c.nc += size; }
      continue;
    }
    if (starts_with(st, "measure")) {
      const size_t arrow = st.find("->");
      if (arrow == View::npos || arrow < 7) throw ParseError{"bad measure '" + head64(st) + "'"};
      scratch.clear();
      const int nqs = bits_of(st.substr(7, arrow - 7), qregs, scratch);
      const int ncs = bits_of(st.substr(arrow + 2), cregs, scratch);
      if (nqs != ncs || nqs == 0) throw ParseError{"measure size mismatch"};
      for (int i = 0; i < nqs; ++i) {
        const int off = (int)c.bits.size();
        c.bits.push_back(scratch[i]); c.bits.push_back(scratch[nqs + i]);
        c.ops.push_back(Op{c.measure, off, 1, off + 1, 1, 0, 0});
      }
      continue;
    }
    if (starts_with(st, "barrier")) {
      const int off = (int)c.bits.size();
      int cnt = 0;
      split_top(st.substr(7), pieces);
      for (const View a : pieces) cnt += bits_of(a, qregs, c.bits);
      c.ops.push_back(Op{c.barrier, off, cnt, off, 0, 0, 0});
      continue;
    }
    if (starts_with(st, "reset")) {
      scratch.clear();
      const int n = bits_of(st.substr(5), qregs, scratch);
      for (int i = 0; i < n; ++i) { c.ops.push_back(Op{c.reset, (int)c.bits.size(), 1, 0, 0, 0, 0}); c.bits.push_back(scratch[i]); }
      continue;
    }
    // name [ (params) ] args
    size_t i = 0;
    while (i < st.size() && is_word(st[i])) ++i;
    if (i == 0) throw ParseError{"cannot parse statement '" + head64(st) + "'"};
    Op op{c.intern(st.substr(0, i)), 0, 0, 0, 0, (int)c.params.size(), 0};
    View rest = trim(st.substr(i));
    if (!rest.empty() && rest[0] == '(') {
      int depth = 0; size_t j = 0;
      for (; j < rest.size(); ++j) { if (rest[j] == '(') ++depth; else if (rest[j] == ')' && --depth == 0) break; }
      if (j >= rest.size()) throw ParseError{"unbalanced parameter list in '" + head64(st) + "'"};
      split_top(rest.substr(1, j - 1), pieces);
      for (const View e : pieces) { c.params.push_back(Expr(e).parse()); ++op.p_cnt; }
      rest = trim(rest.substr(j + 1));
    }
    split_top(rest, pieces);
    if (pieces.empty()) throw ParseError{"statement without qubit arguments '" + head64(st) + "'"};
    scratch.clear(); arg_off.clear(); arg_cnt.clear();
    int width = 1;
    for (const View a : pieces) {
      arg_off.push_back((int)scratch.size());
      arg_cnt.push_back(bits_of(a, qregs, scratch));
      width = std::max(width, arg_cnt.back());
    }
    for (int n : arg_cnt)   // whole-register arguments must agree in size (OpenQASM 2, section 4.2); an empty register has no bit to act on
      if (n == 0 || (n > 1 && n != width)) throw ParseError{"register size mismatch in '" + head64(st) + "'"};
    for (int k = 0; k < width; ++k) {  // whole-register arguments broadcast
      Op o = op;
      o.q_off = (int)c.bits.size(); o.q_cnt = (int)arg_cnt.size(); o.c_off = o.q_off;
      for (size_t a = 0; a < arg_cnt.size(); ++a) c.bits.push_back(scratch[arg_off[a] + (arg_cnt[a] > 1 ? k : 0)]);
      c.ops.push_back(o);
    }
  }
}

Circuit parse_qasm(const char* text) {
  Circuit c;
  std::string s, t;
  parse_qasm(text, c, s, t);
  return c;
}

thread_local std::string g_last_error;

// ---- encoding of a parsed circuit ---------------------------------------------------------------------------------------
struct Sizes { int64_t N = 0, E = 0; int depth = 0; };

// out-lists of the op DAG: for each source op a linked list with the most recently inserted edge at its head -- the order
// the reference's DAG hands its successors back in
struct OutLists {
  std::vector<int> head, next, dst, wire;
  void reset(int64_t n, size_t cap) { head.assign((size_t)n, -1); next.clear(); dst.clear(); wire.clear(); next.reserve(cap); dst.reserve(cap); wire.reserve(cap); }
};

int feature_width(const mlqem_backend_props* props, int use_q, int use_g) { return 3 + props->num_gate_types + 2 + (use_q ? 9 : 0) + (use_g ? 2 : 0); }

// the one-hot column of every interned op name (-1: not in gates_set)
std::vector<int> type_slots(const Circuit& c, const mlqem_backend_props* props) {
  std::vector<int> slot_of(c.names.size(), -1);
  for (size_t t = 0; t < c.names.size(); ++t) {
    for (int i = 0; i < props->num_gate_types; ++i) if (c.names[t] == props->gate_names[i]) slot_of[t] = i;   // the last match, as a dict built in order keeps
    if ((int)t == c.barrier) slot_of[t] = props->num_gate_types;
    if ((int)t == c.measure) slot_of[t] = props->num_gate_types + 1;
  }
  return slot_of;
}

// One pass over the ops in program order: what the encoding refuses, the depth, the edge count (an edge per qubit wire from
// the previous op on it) and, with `lists`, the out-lists (qubit wires first, then clbit wires, per op).
struct WireState { std::vector<int> last, level; };

Sizes scan(const Circuit& c, const mlqem_backend_props* props, int use_q, const std::vector<int>& slot_of, OutLists* lists, WireState& wires) {
  Sizes sz;
  sz.N = (int64_t)c.ops.size();
  std::vector<int>& last = wires.last;
  std::vector<int>& level = wires.level;
  last.assign((size_t)(c.nq + c.nc), -1); level.assign((size_t)(c.nq + c.nc), 0);
  if (lists) lists->reset(sz.N, c.bits.size());
  auto link = [&](int src, int dst, int wire) {
    if (wire < c.nq) ++sz.E;
    if (!lists) return;
    lists->next.push_back(lists->head[src]); lists->dst.push_back(dst); lists->wire.push_back(wire);
    lists->head[src] = (int)lists->dst.size() - 1;
  };
  for (int64_t k = 0; k < sz.N; ++k) {
    const Op& op = c.ops[k];
    const bool is_barrier = op.type == c.barrier;
    if (!is_barrier && op.q_cnt > 3) throw ParseError{"Non barrier gate that has more than 3 qubits.", true};
    if (op.p_cnt > 3) throw ParseError{"more than 3 gate parameters", true};
    // what the fill pass would refuse is refused by the size query too (a caller sizes its buffers, then fills them)
    if (slot_of[op.type] < 0) throw ParseError{"gate '" + c.names[op.type] + "' is not in the backend's gates_set", true};
    const int* qs = c.qubits(op);
    const int* cs = c.clbits(op);
    if (use_q && !is_barrier)
      for (int i = 0; i < op.q_cnt; ++i)
        if (c.reg_index[qs[i]] >= props->num_qubits) throw ParseError{"qubit index beyond the calibration table", true};
    int lvl = 0;
    for (int i = 0; i < op.q_cnt; ++i) { const int q = qs[i]; if (last[q] >= 0) link(last[q], (int)k, q); last[q] = (int)k; lvl = std::max(lvl, level[q]); }
    for (int i = 0; i < op.c_cnt; ++i) { const int w = c.nq + cs[i]; if (last[w] >= 0) link(last[w], (int)k, w); last[w] = (int)k; lvl = std::max(lvl, level[w]); }
    if (!is_barrier) {  // directives do not count towards the depth
      for (int i = 0; i < op.q_cnt; ++i) level[qs[i]] = lvl + 1;
      for (int i = 0; i < op.c_cnt; ++i) level[c.nq + cs[i]] = lvl + 1;
    }
  }
  for (int v : level) sz.depth = std::max(sz.depth, v);
  return sz;
}

// calibration entry of (gate, qubits): the reference's key is "<name>_<q0>[_<q1>...]" (utils.py:139-175); looked up once per
// distinct (name, qubits) through the string, then through a packed integer key (gates on one or two qubits)
struct GateProps {
  std::unordered_map<std::string, int> by_key;
  std::unordered_map<uint64_t, int> packed;
  std::vector<int> one_qubit;            // [type][register-local qubit]: -2 = not looked up yet (gates on one qubit: no hash per node)
  int nq1 = 0;
  std::string key;
  explicit GateProps(const mlqem_backend_props* props) { for (int i = 0; i < props->num_gate_props; ++i) by_key[props->gate_keys[i]] = i; }
  int slow(const Circuit& c, const Op& op) {
    const int* qs = c.qubits(op);
    key = c.names[op.type];
    for (int s = 0; s < op.q_cnt; ++s) { key += '_'; key += std::to_string(c.reg_index[qs[s]]); }
    auto it = by_key.find(key);
    return it == by_key.end() ? -1 : it->second;
  }
  int find(const Circuit& c, const Op& op) {
    const int* qs = c.qubits(op);
    if (op.q_cnt == 1) {
      if (one_qubit.empty()) { nq1 = std::max(c.nq, 1); one_qubit.assign(c.names.size() * (size_t)nq1, -2); }
      const int qi = c.reg_index[qs[0]];
      int& slot = one_qubit[(size_t)op.type * nq1 + qi];     // qi < the register's size <= nq
      if (slot == -2) slot = slow(c, op);
      return slot;
    }
    const bool packable = op.q_cnt <= 2 && op.type < (1 << 20);     // 2 x 21 bits of qubit index, 2 of count, 20 of type
    uint64_t pk = 0;
    if (packable) {
      pk = ((uint64_t)op.type << 44) | ((uint64_t)op.q_cnt << 42);
      for (int s = 0; s < op.q_cnt; ++s) pk |= (uint64_t)c.reg_index[qs[s]] << (21 * s);
      auto hit = packed.find(pk);
      if (hit != packed.end()) return hit->second;
    }
    const int g = slow(c, op);
    if (packable) packed.emplace(pk, g);
    return g;
  }
};

// Feature rows and edges of one circuit (after scan() built `lists`): rows x[N, F] of type XT, edges as (src, dst) of type IT
// with `node_offset` added (a circuit's position in a collated batch), optional edge attributes.
template <typename XT, typename IT>
void fill(const Circuit& c, const mlqem_backend_props* props, int use_q, int use_g, const std::vector<int>& slot_of,
          const OutLists& lists, XT* x, IT* edge_src, IT* edge_dst, IT node_offset, double* edge_attr) {
  const int n_types = props->num_gate_types + 2;  // + barrier, measure
  const int F = feature_width(props, use_q, use_g);
  const int64_t N = (int64_t)c.ops.size();
  GateProps gate_props(props);
  for (int64_t k = 0; k < N; ++k) {
    const Op& op = c.ops[k];
    const int* qs = c.qubits(op);
    XT* row = x + k * F;
    for (int i = 0; i < F; ++i) row[i] = (XT)0;
    for (int i = 0; i < op.p_cnt; ++i) row[i] = (XT)c.params[op.p_off + i];
    row[3 + slot_of[op.type]] = (XT)1;
    int col = 3 + n_types;
    if (use_q) {
      if (op.type != c.barrier)
        for (int s = 0; s < op.q_cnt; ++s) {
          const int qi = c.reg_index[qs[s]];
          row[col + s] = (XT)props->t1[qi]; row[col + 3 + s] = (XT)props->t2[qi]; row[col + 6 + s] = (XT)props->readout[qi];
        }
      col += 9;
    }
    if (use_g) {
      const int g = gate_props.find(c, op);
      if (g >= 0) { row[col] = (XT)props->gate_error[g]; row[col + 1] = (XT)props->gate_length[g]; }
    }
  }
  int64_t e = 0;
  for (int64_t k = 0; k < N; ++k)
    for (int it = lists.head[k]; it >= 0; it = lists.next[it]) {
      const int wire = lists.wire[it];
      if (wire >= c.nq) continue;
      edge_src[e] = (IT)k + node_offset; edge_dst[e] = (IT)lists.dst[it] + node_offset;
      if (edge_attr) {
        const int qi = c.reg_index[wire];
        if (qi >= props->num_qubits) throw ParseError{"qubit index beyond the calibration table", true};
        edge_attr[e * 3] = props->t1[qi]; edge_attr[e * 3 + 1] = props->t2[qi]; edge_attr[e * 3 + 2] = props->readout[qi];
      }
      ++e;
    }
}

int report(const ParseError& err) { g_last_error = err.what; return err.unsupported ? MLQEM_ERR_UNSUPPORTED : MLQEM_ERR_BAD_ARG; }

// ---- a batch of circuits: parsed and sized by one call, filled into the caller's collated buffers by another -------------
struct StreamSizes { int64_t wires = 0, patches = 0; };   // qubit arguments and patch slots of a circuit's op stream (see mlqem_qasm_batch_stream_*)

struct Batch {
  std::vector<Circuit> circuits;
  std::vector<std::vector<int>> slots;
  std::vector<Sizes> sizes;
  std::vector<StreamSizes> stream;          // filled by the parse workers: a serial pass over 11 M ops cost 20 ms of a 30 ms encode
  // circuits / slots / sizes / stream hold ONE entry per DISTINCT text of the run(): entry u_of[i] stands for circuit i.  A run()
  // often names one circuit many times -- the reference's VQE drivers evaluate every Pauli term of a Hamiltonian as its own
  // (circuit, observable) pair of the SAME bound circuit (separate_observables=True, docs/tutorials/vqe_rf.py:245) -- and a text
  // handed over twice as the same buffer is scanned once.  Identity of the pointer, not of the content: no hashing of the texts.
  std::vector<int64_t> u_of;
  int64_t count() const { return (int64_t)u_of.size(); }
  const mlqem_backend_props* props = nullptr;
  int use_q = 0, use_g = 0;
  size_t footprint() const {
    size_t b = 0;
    for (const Circuit& c : circuits) b += c.ops.capacity() * sizeof(Op) + c.bits.capacity() * sizeof(int) + c.params.capacity() * sizeof(double);
    return b;
  }
};

// ONE parsed batch is kept between calls with its circuits' storage (vectors keep their capacity): the next run() of the same
// shape parses straight into it -- no exactly-sized copies, no fresh pages (0.4 GB of them per 1024 100-qubit circuits: their
// page faults serialise on the process's memory map and were what kept 64 threads from scaling).  Bounded: nothing above 768 MB is kept
// (1024 100-qubit circuits parse into 0.4 GB).
// The process-wide pools below hold mutexes, condition variables and (the worker pool) threads.  After fork() the child has ONE thread
// -- the one that forked -- while the pools it inherited still say that workers exist and may hold locks some other parent thread held
// at that instant: a child that encoded again waited for ever on workers that are not there (multiprocessing / DataLoader
// workers started by fork do exactly this).  Every pool therefore sits behind a pointer that a pthread_atfork child handler clears: the
// child abandons the parent's objects (never touches their locks) and builds its own on first use.
template <class T> class ForkSafe {
  static std::atomic<T*>& slot() { static std::atomic<T*> p{nullptr}; return p; }
  static void forget() { slot().store(nullptr, std::memory_order_relaxed); }      // in the child, single-threaded
 public:
  static T& get() {
    T* p = slot().load(std::memory_order_acquire);
    if (!p) {
      static std::once_flag registered;
      std::call_once(registered, [] { pthread_atfork(nullptr, nullptr, &ForkSafe<T>::forget); });
      T* fresh = new T;
      if (slot().compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) p = fresh;
      else delete fresh;                 // another thread was first
    }
    return *p;
  }
};

class BatchPool {
  std::mutex mu_;
  Batch* kept_ = nullptr;
 public:
  Batch* take() {
    std::lock_guard<std::mutex> lock(mu_);
    Batch* b = kept_;
    kept_ = nullptr;
    return b ? b : new Batch;
  }
  void give(Batch* b) {
    if (!b) return;
    if (b->footprint() <= (size_t(768) << 20)) {
      std::lock_guard<std::mutex> lock(mu_);
      if (!kept_) { kept_ = b; return; }
    }
    delete b;
  }
};
BatchPool& batch_pool() { return ForkSafe<BatchPool>::get(); }

// What a worker needs while it scans circuit after circuit.  Kept in a process-wide pool between calls, so that a run() of a
// VQE loop (thousands of calls, blackwater/library/ngem/estimator.py:49-84) does not grow half-megabyte vectors from nothing
// on every worker of every call (each doubling beyond 128 KB is an mmap, a round of page faults and a munmap).
struct WorkerScratch {
  Circuit circuit;
  std::string text_a, text_b;
  OutLists lists;
  WireState wires;
};

class ScratchPool {
  std::mutex mu_;
  std::vector<std::unique_ptr<WorkerScratch>> free_;
 public:
  std::unique_ptr<WorkerScratch> acquire() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (!free_.empty()) { auto p = std::move(free_.back()); free_.pop_back(); return p; }
    }
    return std::make_unique<WorkerScratch>();
  }
  void release(std::unique_ptr<WorkerScratch> p) {
    std::lock_guard<std::mutex> lock(mu_);
    if (free_.size() < 32) free_.push_back(std::move(p));      // a few MB each at most: bounded
  }
};

ScratchPool& scratch_pool() { return ForkSafe<ScratchPool>::get(); }

// Worker threads when the caller names none: the host's hardware threads up to 64 (a text scan is 60 ns per statement per thread,
// and a 1024-circuit run() of 100-qubit circuits has 11 M statements: 16 threads took 42 ms of an 88 ms run()), or
// MLQEM_ENCODE_THREADS (1..256) when the deployment knows better.
// CPUs this process may use at a time by its control group's bandwidth limit (cgroup v2 cpu.max, v1 cfs quota / period), 0 = no
// limit found.  A container that shows 256 cores and grants 16 of them THROTTLES a burst of 64 threads: on such a box a run()'s scan
// took 29 ms and the next one 70 ms, alternately (nr_throttled in cpu.stat counted every second call); at 24 threads every call takes
// 35-40 ms -- a better sustained rate, and the same one every time.
int cgroup_cpu_limit() {
  auto ratio = [](long long quota, long long period) { return (quota > 0 && period > 0) ? (int)((quota + period - 1) / period) : 0; };
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    long long period = 0;
    const int got = std::fscanf(f, "%31s %lld", q, &period);
    std::fclose(f);
    if (got == 2 && std::strcmp(q, "max") != 0) return ratio(std::atoll(q), period);
    if (got >= 1) return 0;
  }
  long long quota = 0, period = 0;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(f, "%lld", &quota) != 1) quota = 0; std::fclose(f); }
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(f, "%lld", &period) != 1) period = 0; std::fclose(f); }
  return ratio(quota, period);
}

int default_threads() {
  if (const char* env = std::getenv("MLQEM_ENCODE_THREADS")) {
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 256) return (int)v;
  }
  static const int chosen = [] {
    int n = (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency()));
    const int limit = cgroup_cpu_limit();
    if (limit > 0) n = std::min(n, std::max(2, limit + limit / 2));      // one and a half threads per granted CPU (measured: 16 / 24 / 32 / 64)
    return n;
  }();
  return chosen;
}

// A process-wide pool of worker threads for the batch entry points.  A run() of a VQE loop calls them thousands of times
// (blackwater/library/ngem/estimator.py:49-84): starting 64 threads per call cost 3-4 ms of a 30 ms encode.  Workers are started
// on demand, wait on a condition variable between jobs and live until the process ends; the pool object is never destroyed (a
// static's destructor would run while workers still wait on its members).  One job at a time (callers queue on `gate`).
class WorkerPool {
  std::mutex mu_, gate_;
  std::condition_variable wake_, done_;
  std::function<void()> job_;
  uint64_t generation_ = 0;
  int wanted_ = 0, running_ = 0;
  std::vector<std::thread> threads_;

  void loop(int index) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        wake_.wait(lock, [&] { return generation_ != seen && index < wanted_; });
        seen = generation_;
        job = job_;
      }
      job();
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (--running_ == 0) done_.notify_all();
      }
    }
  }

 public:
  // runs job() on `helpers` pool threads AND on the calling thread; returns when all have returned.  Fewer helpers than asked
  // for (the system refused a thread) is not an error: the job is written so that any number of runners finishes it.
  void run(int helpers, const std::function<void()>& job) {
    std::lock_guard<std::mutex> one_job(gate_);
    int started = 0;
    {
      std::lock_guard<std::mutex> lock(mu_);
      try {
        while ((int)threads_.size() < helpers) {
          const int index = (int)threads_.size();
          threads_.emplace_back([this, index] { loop(index); });
          threads_.back().detach();
        }
      } catch (const std::exception&) {}
      started = std::min<int>(helpers, (int)threads_.size());
      job_ = job;
      wanted_ = started;
      running_ = started;
      ++generation_;
    }
    if (started > 0) wake_.notify_all();
    job();
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [&] { return running_ == 0; });
    wanted_ = 0;
    job_ = nullptr;
  }
};

WorkerPool& worker_pool() { return ForkSafe<WorkerPool>::get(); }

// fn(i, scratch) for i in [0, count) on `threads` host threads; the first failure (lowest index wins among those seen) is kept
template <typename Fn>
int for_each_parallel(int64_t count, int threads, int64_t* failed, Fn fn) {
  if (threads <= 0) threads = default_threads();
  threads = (int)std::min<int64_t>(threads, std::max<int64_t>(count, 1));
  std::atomic<int64_t> next{0};
  std::mutex mu;
  int code = MLQEM_OK;
  int64_t bad = -1;
  std::string message;
  auto worker = [&]() {
    std::unique_ptr<WorkerScratch> scratch = scratch_pool().acquire();
    struct Return { std::unique_ptr<WorkerScratch>& p; ~Return() { scratch_pool().release(std::move(p)); } } give_back{scratch};
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= count) return;
      int rc = MLQEM_OK;
      std::string what;
      try { fn(i, *scratch); }
      catch (const ParseError& err) { rc = err.unsupported ? MLQEM_ERR_UNSUPPORTED : MLQEM_ERR_BAD_ARG; what = err.what; }
      catch (const std::exception& err) { rc = MLQEM_ERR_BAD_ARG; what = err.what(); }
      if (rc != MLQEM_OK) {
        std::lock_guard<std::mutex> lock(mu);
        if (bad < 0 || i < bad) { bad = i; code = rc; message = what; }
        next.store(count);          // nothing else needs to start
      }
    }
  };
  if (threads == 1) worker();
  else worker_pool().run(threads - 1, worker);       // the caller's thread is one of the workers
  if (code != MLQEM_OK) { g_last_error = "circuit " + std::to_string(bad) + ": " + message; if (failed) *failed = bad; }
  return code;
}

}  // namespace

extern "C" const char* mlqem_encode_last_error(void) { return g_last_error.c_str(); }

extern "C" int mlqem_encode_qasm(const char* qasm, const mlqem_backend_props* props, int use_qubit_features,
                                 int use_gate_features, int64_t* num_nodes, int64_t* num_edges, int* num_features,
                                 int* depth, double* x, int32_t* edge_src, int32_t* edge_dst, double* edge_attr) {
  if (!qasm || !props || !num_nodes || !num_edges || !num_features) return MLQEM_ERR_BAD_ARG;
  try {
    const Circuit c = parse_qasm(qasm);
    const int F = feature_width(props, use_qubit_features, use_gate_features);
    const std::vector<int> slot_of = type_slots(c, props);
    const bool want_fill = x != nullptr;
    OutLists lists;
    WireState wires;
    const Sizes sz = scan(c, props, use_qubit_features, slot_of, want_fill ? &lists : nullptr, wires);
    if (depth) *depth = sz.depth;
    if (want_fill && (*num_nodes < sz.N || *num_edges < sz.E)) {   // says what it needs: the caller may retry with larger buffers
      *num_nodes = sz.N; *num_edges = sz.E; *num_features = F;
      g_last_error = "output buffers too small";
      return MLQEM_ERR_WORKSPACE;
    }
    *num_nodes = sz.N; *num_edges = sz.E; *num_features = F;
    if (!want_fill) return MLQEM_OK;
    if (sz.E > 0 && (!edge_src || !edge_dst)) return MLQEM_ERR_BAD_ARG;
    fill<double, int32_t>(c, props, use_qubit_features, use_gate_features, slot_of, lists, x, edge_src, edge_dst, 0, edge_attr);
    return MLQEM_OK;
  } catch (const ParseError& err) {
    return report(err);
  } catch (const std::exception& err) {   // bad_alloc, length_error: the text asked for more than the host has
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

namespace {

// does any calibration key start with "<name>_"?  (a slot without one never needs a lookup)
std::vector<char> slots_with_calibration(const mlqem_backend_props* props) {
  const int n_slots = props->num_gate_types + 2;
  std::vector<char> any((size_t)n_slots, 0);
  auto name_of = [&](int slot) -> std::string {
    if (slot < props->num_gate_types) return props->gate_names[slot];
    return slot == props->num_gate_types ? "barrier" : "measure";
  };
  for (int slot = 0; slot < n_slots; ++slot) {
    const std::string pre = name_of(slot) + "_";
    for (int i = 0; i < props->num_gate_props && !any[slot]; ++i)
      if (std::strncmp(props->gate_keys[i], pre.c_str(), pre.size()) == 0) any[slot] = 1;
  }
  return any;
}

StreamSizes stream_sizes(const Circuit& c, const std::vector<int>& slot_of, const std::vector<char>& calibrated, int use_g) {
  StreamSizes sz;
  for (const Op& op : c.ops) {
    sz.wires += op.q_cnt;
    if (op.p_cnt > 1) sz.patches += op.p_cnt - 1;
    if (use_g && op.q_cnt > 2 && calibrated[(size_t)slot_of[op.type]]) sz.patches += 2;     // at most: error and length, if the key exists
  }
  return sz;
}

}  // namespace

extern "C" int mlqem_qasm_batch_parse(const char* const* qasm, int64_t count, const mlqem_backend_props* props,
                                      int use_qubit_features, int use_gate_features, int threads, void** handle,
                                      int64_t* node_ptr, int64_t* edge_ptr, int* depths, int* num_features, int64_t* failed) {
  if (!handle) return MLQEM_ERR_BAD_ARG;
  *handle = nullptr;
  if (count < 0 || !props || !node_ptr || !edge_ptr || !num_features || (count > 0 && !qasm)) return MLQEM_ERR_BAD_ARG;
  for (int64_t i = 0; i < count; ++i) if (!qasm[i]) return MLQEM_ERR_BAD_ARG;
  Batch* b = nullptr;
  try {
    b = batch_pool().take();
    // distinct texts (by buffer): first[u] = the first circuit that names text u
    b->u_of.assign((size_t)count, 0);
    std::vector<int64_t> first;
    {
      std::unordered_map<const char*, int64_t> seen;
      seen.reserve((size_t)count * 2 + 1);
      for (int64_t i = 0; i < count; ++i) {
        auto it = seen.find(qasm[i]);
        if (it == seen.end()) { it = seen.emplace(qasm[i], (int64_t)first.size()).first; first.push_back(i); }
        b->u_of[(size_t)i] = it->second;
      }
    }
    const int64_t distinct = (int64_t)first.size();
    b->circuits.resize((size_t)distinct); b->slots.resize((size_t)distinct); b->sizes.resize((size_t)distinct);
    b->props = props; b->use_q = use_qubit_features; b->use_g = use_gate_features;
    b->stream.resize((size_t)distinct);
    const std::vector<char> calibrated = slots_with_calibration(props);
    int64_t failed_u = -1;
    const int rc = for_each_parallel(distinct, threads, &failed_u, [&](int64_t u, WorkerScratch& w) {
      parse_qasm(qasm[first[(size_t)u]], b->circuits[u], w.text_a, w.text_b);       // into the kept circuit: its vectors' capacity is reused
      b->slots[u] = type_slots(b->circuits[u], props);
      b->sizes[u] = scan(b->circuits[u], props, use_qubit_features, b->slots[u], nullptr, w.wires);
      b->stream[u] = stream_sizes(b->circuits[u], b->slots[u], calibrated, use_gate_features);
    });
    if (rc != MLQEM_OK) {
      if (failed_u >= 0) {        // name the circuit by its place in the run(), not among the distinct texts
        const int64_t at = first[(size_t)failed_u];
        if (failed) *failed = at;
        const std::string was = "circuit " + std::to_string(failed_u) + ":";
        if (g_last_error.compare(0, was.size(), was) == 0) g_last_error = "circuit " + std::to_string(at) + ":" + g_last_error.substr(was.size());
      }
      batch_pool().give(b);
      return rc;
    }
    node_ptr[0] = edge_ptr[0] = 0;
    for (int64_t i = 0; i < count; ++i) {
      const Sizes& sz = b->sizes[(size_t)b->u_of[(size_t)i]];
      node_ptr[i + 1] = node_ptr[i] + sz.N;
      edge_ptr[i + 1] = edge_ptr[i] + sz.E;
      if (depths) depths[i] = sz.depth;
    }
    *num_features = feature_width(props, use_qubit_features, use_gate_features);
    *handle = b;
    return MLQEM_OK;
  } catch (const std::exception& err) {
    delete b;
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

extern "C" int mlqem_qasm_batch_fill(void* handle, int threads, float* x, int64_t* edge_src, int64_t* edge_dst, int64_t* batch) {
  Batch* b = static_cast<Batch*>(handle);
  if (!b) { g_last_error = "no batch handle"; return MLQEM_ERR_BAD_ARG; }
  try {                                                        // no C++ exception may cross the C ABI (bad_alloc, system_error)
  const int64_t count = b->count();
  std::vector<int64_t> node_ptr((size_t)count + 1, 0), edge_ptr((size_t)count + 1, 0);
  for (int64_t i = 0; i < count; ++i) {
    const Sizes& sz = b->sizes[(size_t)b->u_of[(size_t)i]];
    node_ptr[i + 1] = node_ptr[i] + sz.N; edge_ptr[i + 1] = edge_ptr[i] + sz.E;
  }
  if ((node_ptr[count] > 0 && !x) || (edge_ptr[count] > 0 && (!edge_src || !edge_dst))) {   // an empty batch needs no buffers
    g_last_error = "missing output buffer";
    return MLQEM_ERR_BAD_ARG;
  }
  const int F = feature_width(b->props, b->use_q, b->use_g);
  return for_each_parallel(count, threads, nullptr, [&](int64_t i, WorkerScratch& w) {
    const int64_t u = b->u_of[(size_t)i];
    const Circuit& c = b->circuits[u];
    scan(c, b->props, b->use_q, b->slots[u], &w.lists, w.wires);
    fill<float, int64_t>(c, b->props, b->use_q, b->use_g, b->slots[u], w.lists, x + node_ptr[i] * F, edge_src + edge_ptr[i],
                         edge_dst + edge_ptr[i], node_ptr[i], nullptr);
    if (batch) std::fill(batch + node_ptr[i], batch + node_ptr[i + 1], i);
  });
  } catch (const std::exception& err) {
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

// ---- the compact op stream of a parsed batch (device-side expansion: csrc/encode_expand.hip) ----------------------------
// What the GPU needs to rebuild x, edge_index and batch is 16 bytes per op (first parameter, up to three calibration qubit
// indices, one-hot column, counts) plus two bytes per qubit argument (the wire, for the op -> op edges) -- 0.2 GB for a
// 1024-circuit run() of 100-qubit circuits instead of the 1.3 GB of float32 rows and int64 indices mlqem_qasm_batch_fill writes.
// Values the record has no room for (second / third parameters, calibration entries of three-qubit gates) travel as patches.
namespace {

}  // namespace

extern "C" int mlqem_qasm_batch_stream_sizes(void* handle, int64_t* wire_ptr, int64_t* patch_ptr, int* max_wires) {
  Batch* b = static_cast<Batch*>(handle);
  if (!b || !wire_ptr || !patch_ptr) { g_last_error = "no batch handle or output"; return MLQEM_ERR_BAD_ARG; }
  try {
    const int64_t count = b->count();
    wire_ptr[0] = patch_ptr[0] = 0;
    int widest = 0;
    for (int64_t i = 0; i < count; ++i) {
      const int64_t u = b->u_of[(size_t)i];
      const Circuit& c = b->circuits[u];
      if (c.nq > 65535 || b->props->num_qubits > 65535) { g_last_error = "more than 65535 wires: the op stream holds 16-bit indices"; return MLQEM_ERR_UNSUPPORTED; }
      const StreamSizes sz = b->stream[u];
      wire_ptr[i + 1] = wire_ptr[i] + sz.wires;
      patch_ptr[i + 1] = patch_ptr[i] + sz.patches;
      widest = std::max(widest, c.nq);
    }
    if (max_wires) *max_wires = widest;
    return MLQEM_OK;
  } catch (const std::exception& err) {
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

// ops[node_ptr[count]], wires[wire_ptr[count]], patches[patch_ptr[count]] (capacity; *num_patches = how many were written, packed
// at the front circuit by circuit is NOT guaranteed: unused slots hold node = 0xFFFFFFFF and are skipped by the device)
extern "C" int mlqem_qasm_batch_stream_fill(void* handle, int threads, const int64_t* wire_ptr, const int64_t* patch_ptr,
                                            mlqem_op_rec* ops, uint16_t* wires, mlqem_x_patch* patches) {
  Batch* b = static_cast<Batch*>(handle);
  if (!b || !wire_ptr || !patch_ptr) { g_last_error = "no batch handle"; return MLQEM_ERR_BAD_ARG; }
  try {
    const int64_t count = b->count();
    std::vector<int64_t> node_ptr((size_t)count + 1, 0);
    for (int64_t i = 0; i < count; ++i) node_ptr[i + 1] = node_ptr[i] + b->sizes[(size_t)b->u_of[(size_t)i]].N;
    if ((node_ptr[count] > 0 && !ops) || (wire_ptr[count] > 0 && !wires) || (patch_ptr[count] > 0 && !patches)) {
      g_last_error = "missing output buffer";
      return MLQEM_ERR_BAD_ARG;
    }
    if (wire_ptr[count] >= (1ll << 32) || node_ptr[count] >= (1ll << 32)) { g_last_error = "op stream beyond 2^32 entries"; return MLQEM_ERR_UNSUPPORTED; }
    const std::vector<char> calibrated = slots_with_calibration(b->props);
    const int F = feature_width(b->props, b->use_q, b->use_g);
    return for_each_parallel(count, threads, nullptr, [&](int64_t i, WorkerScratch&) {
      const int64_t u = b->u_of[(size_t)i];
      const Circuit& c = b->circuits[u];
      const std::vector<int>& slot_of = b->slots[u];
      mlqem_op_rec* out = ops + node_ptr[i];
      uint16_t* w = wires + wire_ptr[i];
      mlqem_x_patch* px = patches + patch_ptr[i];
      mlqem_x_patch* const px_end = patches + patch_ptr[i + 1];
      uint32_t winc = (uint32_t)wire_ptr[i];
      GateProps gate_props(b->props);
      const int64_t N = (int64_t)c.ops.size();
      for (int64_t k = 0; k < N; ++k) {
        const Op& op = c.ops[k];
        const int* qs = c.qubits(op);
        const bool is_barrier = op.type == c.barrier;
        mlqem_op_rec r;
        r.p0 = op.p_cnt > 0 ? (float)c.params[op.p_off] : 0.f;
        r.q[0] = r.q[1] = r.q[2] = 0;
        if (!is_barrier)
          for (int s2 = 0; s2 < op.q_cnt && s2 < 3; ++s2) r.q[s2] = (uint16_t)std::min(c.reg_index[qs[s2]], 65535);
        r.slot = (uint8_t)slot_of[op.type];
        r.meta = (uint8_t)((is_barrier ? 0 : std::min(op.q_cnt, 3)) | (is_barrier ? 4 : 0) | (std::min(op.p_cnt, 3) << 4));
        r.winc = winc;
        out[k] = r;
        for (int s2 = 0; s2 < op.q_cnt; ++s2) w[s2] = (uint16_t)qs[s2];
        w += op.q_cnt;
        winc += (uint32_t)op.q_cnt;
        const uint32_t node = (uint32_t)(node_ptr[i] + k);
        for (int s2 = 1; s2 < op.p_cnt; ++s2) *px++ = mlqem_x_patch{node, (uint32_t)s2, (float)c.params[op.p_off + s2]};
        if (b->use_g && op.q_cnt > 2 && calibrated[(size_t)slot_of[op.type]]) {
          const int g = gate_props.find(c, op);
          if (g >= 0) {
            *px++ = mlqem_x_patch{node, (uint32_t)(F - 2), (float)b->props->gate_error[g]};
            *px++ = mlqem_x_patch{node, (uint32_t)(F - 1), (float)b->props->gate_length[g]};
          }
        }
      }
      for (; px < px_end; ++px) *px = mlqem_x_patch{0xFFFFFFFFu, 0u, 0.f};
    });
  } catch (const std::exception& err) {
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

// Calibration tables for the device: g1[slot * nq + q] / g2[(slot * nq + q0) * nq + q1] = index of the entry "<name>_<q>" /
// "<name>_<q0>_<q1>" (-1: none), where <name> is the slot's op name -- the keys the reference builds per node (utils.py:263-269).
// Every key is matched against every slot name, so a name that itself ends in "_<digits>" is told apart as the string lookup does.
extern "C" int mlqem_props_gate_tables(const mlqem_backend_props* props, int32_t* g1, int32_t* g2) {
  if (!props || !g1 || !g2) return MLQEM_ERR_BAD_ARG;
  try {
    const int nq = props->num_qubits, n_slots = props->num_gate_types + 2;
    for (int64_t i = 0; i < (int64_t)n_slots * nq; ++i) g1[i] = -1;
    for (int64_t i = 0; i < (int64_t)n_slots * nq * nq; ++i) g2[i] = -1;
    auto number = [](const char* p, const char** end) -> long {     // decimal digits (no sign, no leading zeros but "0"), -1 otherwise
      if (!is_digit(*p) || (*p == '0' && is_digit(p[1]))) return -1;
      long v = 0;
      while (is_digit(*p)) { v = v * 10 + (*p - '0'); if (v > 1000000) return -1; ++p; }
      *end = p;
      return v;
    };
    for (int i = 0; i < props->num_gate_props; ++i) {           // in order: a later duplicate key replaces an earlier one, as in a dict
      const char* key = props->gate_keys[i];
      for (int slot = 0; slot < n_slots; ++slot) {
        const std::string name = slot < props->num_gate_types ? std::string(props->gate_names[slot]) : (slot == props->num_gate_types ? "barrier" : "measure");
        if (std::strncmp(key, name.c_str(), name.size()) != 0 || key[name.size()] != '_') continue;
        const char* p = key + name.size() + 1;
        const char* e = p;
        const long q0 = number(p, &e);
        if (q0 < 0) continue;
        if (*e == '\0') { if (q0 < nq) g1[(int64_t)slot * nq + q0] = i; continue; }
        if (*e != '_') continue;
        const char* e2 = e + 1;
        const long q1 = number(e + 1, &e2);
        if (q1 < 0 || *e2 != '\0') continue;
        if (q0 < nq && q1 < nq) g2[((int64_t)slot * nq + q0) * nq + q1] = i;
      }
    }
    return MLQEM_OK;
  } catch (const std::exception& err) {
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

extern "C" void mlqem_qasm_batch_free(void* handle) { batch_pool().give(static_cast<Batch*>(handle)); }

// Circuit-level features of the MLP regressors (docs/tutorials/mlp.py:111-145, 148-252): by-products of the same op scan.
namespace {

void circuit_features(const Circuit& c, const char* const* gate_names, int num_gates, const double* bin_edges, int num_edges,
                      int64_t* gate_counts, int64_t* angle_hist) {
  for (int i = 0; i < num_gates; ++i) gate_counts[i] = 0;
  const int bins = num_edges > 1 ? num_edges - 1 : 0;
  for (int i = 0; i < bins; ++i) angle_hist[i] = 0;
  std::vector<int64_t> per_type(c.names.size(), 0);
  std::vector<char> rot(c.names.size(), 0);
  for (size_t t = 0; t < c.names.size(); ++t) rot[t] = c.names[t] == "rx" || c.names[t] == "ry" || c.names[t] == "rz";
  for (const Op& op : c.ops) {
    ++per_type[op.type];
    if (!rot[op.type] || op.q_cnt != 1 || op.p_cnt == 0 || bins == 0) continue;
    const double a = c.params[op.p_off];
    if (!(a >= bin_edges[0]) || a > bin_edges[bins]) continue;          // outside, or NaN
    // numpy.histogram with explicit edges: [e_i, e_{i+1}) and a closed last bin
    int b = (int)(std::upper_bound(bin_edges, bin_edges + num_edges, a) - bin_edges) - 1;
    if (b >= bins) b = bins - 1;
    ++angle_hist[b];
  }
  for (int i = 0; i < num_gates; ++i)          // a name listed twice is counted under both entries
    for (size_t t = 0; t < c.names.size(); ++t) if (c.names[t] == gate_names[i]) gate_counts[i] += per_type[t];
}

bool features_args_ok(int num_gates, int num_edges, const char* const* gate_names, const double* bin_edges, const int64_t* gate_counts,
                      const int64_t* angle_hist) {
  return num_gates >= 0 && num_edges >= 0 && !(num_gates && (!gate_names || !gate_counts)) && !(num_edges > 1 && (!bin_edges || !angle_hist));
}

}  // namespace

extern "C" int mlqem_circuit_features_qasm(const char* qasm, const char* const* gate_names, int num_gates,
                                           const double* bin_edges, int num_edges, int64_t* gate_counts,
                                           int64_t* angle_hist) {
  if (!qasm || !features_args_ok(num_gates, num_edges, gate_names, bin_edges, gate_counts, angle_hist) || (num_edges && (!bin_edges || !angle_hist)))
    return MLQEM_ERR_BAD_ARG;
  try {
    const Circuit c = parse_qasm(qasm);
    circuit_features(c, gate_names, num_gates, bin_edges, num_edges, gate_counts, angle_hist);
    return MLQEM_OK;
  } catch (const ParseError& e) {
    return report(e);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

extern "C" int mlqem_circuit_features_qasm_batch(const char* const* qasm, int64_t count, const char* const* gate_names, int num_gates,
                                                 const double* bin_edges, int num_edges, int threads, int64_t* gate_counts,
                                                 int64_t* angle_hist, int64_t* failed) {
  if (count < 0 || (count > 0 && !qasm) || !features_args_ok(num_gates, num_edges, gate_names, bin_edges, gate_counts, angle_hist) ||
      (num_edges && (!bin_edges || !angle_hist)))
    return MLQEM_ERR_BAD_ARG;
  for (int64_t i = 0; i < count; ++i) if (!qasm[i]) return MLQEM_ERR_BAD_ARG;
  const int bins = num_edges > 1 ? num_edges - 1 : 0;
  try {
    return for_each_parallel(count, threads, failed, [&](int64_t i, WorkerScratch& w) {
      parse_qasm(qasm[i], w.circuit, w.text_a, w.text_b);
      circuit_features(w.circuit, gate_names, num_gates, bin_edges, num_edges, gate_counts + i * num_gates, angle_hist + i * bins);
    });
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return MLQEM_ERR_BAD_ARG;
  }
}
