// Host-side (CPU) native encoder: OpenQASM-2 text -> the op-node feature matrix and the op->op qubit-wire edge list
// of the reference's circuit_to_graph_data_json (blackwater/data/utils.py:198-389), i.e. exactly what
// ExpValueEntry.to_pyg_data consumes (blackwater/data/generators/exp_val.py:63-70).  Same node order (program order),
// same feature layout, same edge order (for each source op, its out-edges most recently inserted first), same doubles.
// No DAG library: a DAG built by appending ops has one chain per wire, so the edge list follows from "the previous op
// on each wire".  Pure C++17, no HIP calls: it can run on a box without a GPU.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mlqem_hip.h"

namespace {

struct Op {
  std::string name;
  std::vector<int> qubits, clbits;   // flat indices
  std::vector<double> params;
};

struct Reg { int base, size; };

// Two kinds of failure: text that is not OpenQASM 2 (MLQEM_ERR_BAD_ARG) and a well-formed circuit the encoding does not
// cover (a gate outside gates_set, more than 3 qubits / parameters: MLQEM_ERR_UNSUPPORTED, where the reference raises too).
struct ParseError { std::string what; bool unsupported = false; };

// Limits that keep hostile text from turning into unbounded recursion or allocation: the text arrives through a public
// entry point (the decorators hand over whatever the caller's circuits serialise to).
constexpr int kMaxExprDepth = 64;           // nested parentheses / function calls / unary signs in one angle expression
constexpr long kMaxRegisterBits = 1 << 20;  // qubits + clbits of a circuit (the largest devices have ~1e3)

// decimal integer followed by ']' at s (after optional spaces); -1 when it is anything else or does not fit
long bracket_index(const char* s) {
  while (*s && std::isspace((unsigned char)*s)) ++s;
  if (!std::isdigit((unsigned char)*s)) return -1;
  long v = 0;
  for (; std::isdigit((unsigned char)*s); ++s) {
    v = v * 10 + (*s - '0');
    if (v > kMaxRegisterBits) return -1;
  }
  while (*s && std::isspace((unsigned char)*s)) ++s;
  return *s == ']' ? v : -1;
}

// ---- angle expressions: numbers, pi, + - * / ^, unary sign, parentheses, sin cos tan exp ln sqrt asin acos atan
struct Expr {
  const char* p;
  int depth = 0;
  struct Nest {   // one level of recursion (a parenthesis, a function argument, a unary sign, an exponent)
    int& d;
    explicit Nest(int& depth) : d(depth) { if (++d > kMaxExprDepth) throw ParseError{"angle expression nested too deeply"}; }
    ~Nest() { --d; }
  };
  explicit Expr(const char* s) : p(s) {}
  void ws() { while (*p && std::isspace((unsigned char)*p)) ++p; }
  double parse() { double v = sum(); ws(); if (*p) throw ParseError{"trailing characters in angle expression"}; return v; }
  double sum() {
    double v = product();
    for (;;) {
      ws();
      if (*p == '+') { ++p; v = v + product(); }
      else if (*p == '-') { ++p; v = v - product(); }
      else return v;
    }
  }
  double product() {
    double v = unary();
    for (;;) {
      ws();
      if (*p == '*') { ++p; v = v * unary(); }
      else if (*p == '/') { ++p; v = v / unary(); }
      else return v;
    }
  }
  double unary() {
    bool neg = false;                       // a run of signs is a loop, not a recursion
    for (ws(); *p == '-' || *p == '+'; ws()) { if (*p == '-') neg = !neg; ++p; }
    const double v = power();
    return neg ? -v : v;
  }
  double power() {
    double b = atom();
    ws();
    if (*p == '^') { ++p; Nest n(depth); return std::pow(b, unary()); }
    return b;
  }
  double atom() {
    ws();
    if (*p == '(') { Nest n(depth); ++p; double v = sum(); ws(); if (*p != ')') throw ParseError{"missing ) in angle expression"}; ++p; return v; }
    if (std::isdigit((unsigned char)*p) || *p == '.') { char* end; double v = std::strtod(p, &end); p = end; return v; }
    if (std::isalpha((unsigned char)*p)) {
      std::string id;
      while (std::isalnum((unsigned char)*p) || *p == '_') id.push_back(*p++);
      if (id == "pi") return M_PI;
      ws();
      if (*p != '(') throw ParseError{"unknown identifier '" + id + "' in angle expression"};
      Nest n(depth);
      ++p; double a = sum(); ws(); if (*p != ')') throw ParseError{"missing ) after function"}; ++p;
      if (id == "sin") return std::sin(a); if (id == "cos") return std::cos(a); if (id == "tan") return std::tan(a);
      if (id == "exp") return std::exp(a); if (id == "ln") return std::log(a); if (id == "sqrt") return std::sqrt(a);
      if (id == "asin") return std::asin(a); if (id == "acos") return std::acos(a); if (id == "atan") return std::atan(a);
      throw ParseError{"unknown function '" + id + "'"};
    }
    throw ParseError{"bad angle expression"};
  }
};

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

std::vector<std::string> split_top(const std::string& s) {  // commas outside parentheses
  std::vector<std::string> out; std::string cur; int depth = 0;
  for (char c : s) {
    if (c == '(') ++depth; else if (c == ')') --depth;
    if (c == ',' && depth == 0) { out.push_back(trim(cur)); cur.clear(); } else cur.push_back(c);
  }
  if (!trim(cur).empty() || !out.empty()) out.push_back(trim(cur));   // "a," has an (empty) second element: the callers reject it
  return out;
}

bool starts_with(const std::string& s, const char* pre) { return s.compare(0, std::strlen(pre), pre) == 0; }

struct Circuit {
  int nq = 0, nc = 0;
  std::vector<int> reg_index;            // register-local index of every flat qubit (qiskit's Qubit.index)
  std::vector<Op> ops;
};

std::vector<int> bits_of(const std::string& arg, const std::unordered_map<std::string, Reg>& regs) {
  std::string a = trim(arg);
  size_t br = a.find('[');
  std::string name = trim(br == std::string::npos ? a : a.substr(0, br));
  auto it = regs.find(name);
  if (it == regs.end()) throw ParseError{"unknown register in '" + a + "'"};
  std::vector<int> out;
  if (br == std::string::npos) { for (int i = 0; i < it->second.size; ++i) out.push_back(it->second.base + i); return out; }
  const long idx = bracket_index(a.c_str() + br + 1);
  if (idx < 0 || idx >= it->second.size) throw ParseError{"bad or out-of-range index in '" + a + "'"};
  out.push_back(it->second.base + (int)idx);
  return out;
}

Circuit parse_qasm(const char* text) {
  std::string src(text);
  // strip // comments
  std::string s; s.reserve(src.size());
  for (size_t i = 0; i < src.size(); ++i) {
    if (src[i] == '/' && i + 1 < src.size() && src[i + 1] == '/') { while (i < src.size() && src[i] != '\n') ++i; }
    if (i < src.size()) s.push_back(src[i]);
  }
  // drop gate / opaque definitions (kept opaque: an op using one carries the definition's name)
  std::string t; t.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    bool at_word = (i == 0 || !(std::isalnum((unsigned char)s[i - 1]) || s[i - 1] == '_'));
    if (at_word && (s.compare(i, 5, "gate ") == 0 || s.compare(i, 7, "opaque ") == 0)) {
      size_t brace = s.find('{', i), semi = s.find(';', i);
      if (brace != std::string::npos && (semi == std::string::npos || brace < semi)) {
        size_t close = s.find('}', brace);
        if (close == std::string::npos) throw ParseError{"unterminated gate definition"};
        i = close + 1;
      } else {
        if (semi == std::string::npos) throw ParseError{"unterminated opaque declaration"};
        i = semi + 1;
      }
      continue;
    }
    t.push_back(s[i++]);
  }
  Circuit c;
  std::unordered_map<std::string, Reg> qregs, cregs;
  size_t pos = 0;
  while (pos < t.size()) {
    size_t semi = t.find(';', pos);
    std::string st = trim(t.substr(pos, semi == std::string::npos ? std::string::npos : semi - pos));
    pos = semi == std::string::npos ? t.size() : semi + 1;
    if (st.empty() || starts_with(st, "OPENQASM") || starts_with(st, "include")) continue;
    if (starts_with(st, "qreg") || starts_with(st, "creg")) {
      bool q = st[0] == 'q';
      std::string rest = trim(st.substr(4));
      size_t br = rest.find('[');
      if (br == std::string::npos) throw ParseError{"bad register declaration '" + st + "'"};
      std::string name = trim(rest.substr(0, br));
      const long size_l = bracket_index(rest.c_str() + br + 1);
      if (name.empty() || size_l < 0 || c.nq + c.nc + size_l > kMaxRegisterBits) throw ParseError{"bad register declaration '" + st + "'"};
      if (qregs.count(name) || cregs.count(name)) throw ParseError{"register '" + name + "' declared twice"};
      const int size = (int)size_l;
      if (q) { qregs[name] = Reg{c.nq, size}; c.nq += size; for (int i = 0; i < size; ++i) c.reg_index.push_back(i); }
      else { cregs[name] = Reg{c.nc, size}; c.nc += size; }
      continue;
    }
    if (starts_with(st, "measure")) {
      size_t arrow = st.find("->");
      if (arrow == std::string::npos) throw ParseError{"bad measure '" + st + "'"};
      auto qs = bits_of(st.substr(7, arrow - 7), qregs);
      auto cs = bits_of(st.substr(arrow + 2), cregs);
      if (qs.size() != cs.size() || qs.empty()) throw ParseError{"measure size mismatch"};
      for (size_t i = 0; i < qs.size(); ++i) c.ops.push_back(Op{"measure", {qs[i]}, {cs[i]}, {}});
      continue;
    }
    if (starts_with(st, "barrier")) {
      Op op{"barrier", {}, {}, {}};
      for (auto& a : split_top(st.substr(7))) for (int b : bits_of(a, qregs)) op.qubits.push_back(b);
      c.ops.push_back(op);
      continue;
    }
    if (starts_with(st, "reset")) {
      for (int b : bits_of(st.substr(5), qregs)) c.ops.push_back(Op{"reset", {b}, {}, {}});
      continue;
    }
    // name [ (params) ] args
    size_t i = 0;
    while (i < st.size() && (std::isalnum((unsigned char)st[i]) || st[i] == '_')) ++i;
    if (i == 0) throw ParseError{"cannot parse statement '" + st.substr(0, 64) + "'"};
    Op op; op.name = st.substr(0, i);
    std::string rest = trim(st.substr(i));
    if (!rest.empty() && rest[0] == '(') {
      int depth = 0; size_t j = 0;
      for (; j < rest.size(); ++j) { if (rest[j] == '(') ++depth; else if (rest[j] == ')' && --depth == 0) break; }
      if (j >= rest.size()) throw ParseError{"unbalanced parameter list in '" + st.substr(0, 64) + "'"};
      for (auto& e : split_top(rest.substr(1, j - 1))) op.params.push_back(Expr(e.c_str()).parse());
      rest = trim(rest.substr(j + 1));
    }
    std::vector<std::vector<int>> args;
    size_t width = 1;
    for (auto& a : split_top(rest)) { args.push_back(bits_of(a, qregs)); width = std::max(width, args.back().size()); }
    if (args.empty()) throw ParseError{"statement without qubit arguments '" + st.substr(0, 64) + "'"};
    for (auto& a : args)   // whole-register arguments must agree in size (OpenQASM 2, section 4.2); an empty register has no bit to act on
      if (a.empty() || (a.size() > 1 && a.size() != width)) throw ParseError{"register size mismatch in '" + st.substr(0, 64) + "'"};
    for (size_t k = 0; k < width; ++k) {  // whole-register arguments broadcast
      Op o = op;
      for (auto& a : args) o.qubits.push_back(a.size() > 1 ? a[k] : a[0]);
      c.ops.push_back(o);
    }
  }
  return c;
}

thread_local std::string g_last_error;

}  // namespace

extern "C" const char* mlqem_encode_last_error(void) { return g_last_error.c_str(); }

extern "C" int mlqem_encode_qasm(const char* qasm, const mlqem_backend_props* props, int use_qubit_features,
                                 int use_gate_features, int64_t* num_nodes, int64_t* num_edges, int* num_features,
                                 int* depth, double* x, int32_t* edge_src, int32_t* edge_dst, double* edge_attr) {
  if (!qasm || !props || !num_nodes || !num_edges || !num_features) return MLQEM_ERR_BAD_ARG;
  try {
    const Circuit c = parse_qasm(qasm);
    const int n_types = props->num_gate_types + 2;  // + barrier, measure
    const int F = 3 + n_types + (use_qubit_features ? 9 : 0) + (use_gate_features ? 2 : 0);
    std::unordered_map<std::string, int> type_slot;
    for (int i = 0; i < props->num_gate_types; ++i) type_slot[props->gate_names[i]] = i;
    type_slot["barrier"] = props->num_gate_types;
    type_slot["measure"] = props->num_gate_types + 1;
    std::unordered_map<std::string, int> gate_prop;
    for (int i = 0; i < props->num_gate_props; ++i) gate_prop[props->gate_keys[i]] = i;

    // edges: previous op on each wire (qubits first, then clbits), out-lists in insertion order
    const int64_t N = (int64_t)c.ops.size();
    std::vector<int> last(c.nq + c.nc, -1);
    std::vector<std::vector<std::pair<int, int>>> out(N);  // (dst, wire)
    std::vector<int> level(c.nq + c.nc, 0);
    for (int64_t k = 0; k < N; ++k) {
      const Op& op = c.ops[k];
      if (op.name != "barrier" && op.qubits.size() > 3) throw ParseError{"Non barrier gate that has more than 3 qubits.", true};
      if (op.params.size() > 3) throw ParseError{"more than 3 gate parameters", true};
      // what the fill pass would refuse is refused by the size query too (a caller sizes its buffers, then fills them)
      if (!type_slot.count(op.name)) throw ParseError{"gate '" + op.name + "' is not in the backend's gates_set", true};
      if (use_qubit_features && op.name != "barrier")
        for (int q : op.qubits)
          if (c.reg_index[q] >= props->num_qubits) throw ParseError{"qubit index beyond the calibration table", true};
      int lvl = 0;
      for (int q : op.qubits) { if (last[q] >= 0) out[last[q]].push_back({(int)k, q}); last[q] = (int)k; lvl = std::max(lvl, level[q]); }
      for (int cb : op.clbits) { int w = c.nq + cb; if (last[w] >= 0) out[last[w]].push_back({(int)k, w}); last[w] = (int)k; lvl = std::max(lvl, level[w]); }
      if (op.name != "barrier") {  // directives do not count towards the depth
        for (int q : op.qubits) level[q] = lvl + 1;
        for (int cb : op.clbits) level[c.nq + cb] = lvl + 1;
      }
    }
    int64_t E = 0;
    for (int64_t k = 0; k < N; ++k) for (auto& e : out[k]) if (e.second < c.nq) ++E;
    if (depth) { int d = 0; for (int v : level) d = std::max(d, v); *depth = d; }
    const bool fill = x != nullptr;
    if (fill && (*num_nodes < N || *num_edges < E)) { g_last_error = "output buffers too small"; return MLQEM_ERR_WORKSPACE; }
    *num_nodes = N; *num_edges = E; *num_features = F;
    if (!fill) return MLQEM_OK;
    if (E > 0 && (!edge_src || !edge_dst)) return MLQEM_ERR_BAD_ARG;

    for (int64_t k = 0; k < N; ++k) {
      const Op& op = c.ops[k];
      double* row = x + k * F;
      for (int i = 0; i < F; ++i) row[i] = 0.0;
      for (size_t i = 0; i < op.params.size(); ++i) row[i] = op.params[i];
      auto it = type_slot.find(op.name);
      if (it == type_slot.end()) throw ParseError{"gate '" + op.name + "' is not in the backend's gates_set", true};
      row[3 + it->second] = 1.0;
      int col = 3 + n_types;
      if (use_qubit_features) {
        if (op.name != "barrier")
          for (size_t s = 0; s < op.qubits.size(); ++s) {
            const int qi = c.reg_index[op.qubits[s]];
            if (qi >= props->num_qubits) throw ParseError{"qubit index beyond the calibration table", true};
            row[col + s] = props->t1[qi]; row[col + 3 + s] = props->t2[qi]; row[col + 6 + s] = props->readout[qi];
          }
        col += 9;
      }
      if (use_gate_features) {
        std::string key = op.name;
        for (int q : op.qubits) key += "_" + std::to_string(c.reg_index[q]);
        auto g = gate_prop.find(key);
        if (g != gate_prop.end()) { row[col] = props->gate_error[g->second]; row[col + 1] = props->gate_length[g->second]; }
      }
    }
    int64_t e = 0;
    for (int64_t k = 0; k < N; ++k)
      for (auto it = out[k].rbegin(); it != out[k].rend(); ++it) {
        if (it->second >= c.nq) continue;
        edge_src[e] = (int32_t)k; edge_dst[e] = it->first;
        if (edge_attr) {
          const int qi = c.reg_index[it->second];
          if (qi >= props->num_qubits) throw ParseError{"qubit index beyond the calibration table", true};
          edge_attr[e * 3] = props->t1[qi]; edge_attr[e * 3 + 1] = props->t2[qi]; edge_attr[e * 3 + 2] = props->readout[qi];
        }
        ++e;
      }
    return MLQEM_OK;
  } catch (const ParseError& err) {
    g_last_error = err.what;
    return err.unsupported ? MLQEM_ERR_UNSUPPORTED : MLQEM_ERR_BAD_ARG;
  } catch (const std::exception& err) {   // bad_alloc, length_error: the text asked for more than the host has
    g_last_error = err.what();
    return MLQEM_ERR_BAD_ARG;
  }
}

// Circuit-level features of the MLP regressors (docs/tutorials/mlp.py:111-145, 148-252): by-products of the same op scan.
extern "C" int mlqem_circuit_features_qasm(const char* qasm, const char* const* gate_names, int num_gates,
                                           const double* bin_edges, int num_edges, int64_t* gate_counts,
                                           int64_t* angle_hist) {
  if (!qasm || num_gates < 0 || num_edges < 0 || (num_gates && (!gate_names || !gate_counts)) ||
      (num_edges && (!bin_edges || !angle_hist)))
    return MLQEM_ERR_BAD_ARG;
  try {
    const Circuit c = parse_qasm(qasm);
    for (int i = 0; i < num_gates; ++i) gate_counts[i] = 0;
    const int bins = num_edges > 1 ? num_edges - 1 : 0;
    for (int i = 0; i < bins; ++i) angle_hist[i] = 0;
    for (const Op& op : c.ops) {
      for (int i = 0; i < num_gates; ++i) if (op.name == gate_names[i]) ++gate_counts[i];
      const bool rot = op.name == "rx" || op.name == "ry" || op.name == "rz";
      if (!rot || op.qubits.size() != 1 || op.params.empty() || bins == 0) continue;
      const double a = op.params[0];
      if (!(a >= bin_edges[0]) || a > bin_edges[bins]) continue;          // outside, or NaN
      // numpy.histogram with explicit edges: [e_i, e_{i+1}) and a closed last bin
      int b = (int)(std::upper_bound(bin_edges, bin_edges + num_edges, a) - bin_edges) - 1;
      if (b >= bins) b = bins - 1;
      ++angle_hist[b];
    }
    return MLQEM_OK;
  } catch (const ParseError& e) {
    g_last_error = e.what;
    return e.unsupported ? MLQEM_ERR_UNSUPPORTED : MLQEM_ERR_BAD_ARG;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return MLQEM_ERR_BAD_ARG;
  }
}
