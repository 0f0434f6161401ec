// Behavioural contract: reference tools/mps_load.py:7-89 on top of pulp's
// LpProblem.fromMPS (whitespace-split fields; default bounds [0, +inf);
// problem variables sorted by name when the bound lines are written).
#include "mps_text.h"

#include "netlib_text.h"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace {

struct Bound {
  bool has_low = true, has_up = false;
  double low = 0.0, up = 0.0;
};
typedef std::vector<std::pair<std::string, double>> Terms;

std::string F(double x) {   // Python's format(x, "f")
  char buf[400];
  std::snprintf(buf, sizeof(buf), "%f", x);
  return buf;
}

// mps_load.py:18-34: negative coefficients bring their own sign, the others are joined with '+'.
std::string WriteExpression(const Terms& terms, double constant, const std::map<std::string, std::string>& rename) {
  std::string out;
  for (size_t k = 0; k < terms.size(); k++) {
    if (!(terms[k].second < 0) && k > 0) out += "+";
    out += F(terms[k].second) + "*" + rename.at(terms[k].first);
  }
  if (!terms.empty()) out += "+";
  return out + F(constant);
}

}  // namespace

std::string MpsToLpText(const std::string& mps_path) {
  std::ifstream in(mps_path);
  if (!in.is_open()) throw std::runtime_error("Cannot find file\n");
  if (LooksLikeCompressedNetlib(in)) {   // netlib's compressed form: expand it first (tools/preprocessing.sh:2-9)
    std::istringstream mps(ExpandNetlib(in, mps_path));
    return MpsStreamToLpText(mps);
  }
  return MpsStreamToLpText(in);
}

std::string MpsStreamToLpText(std::istream& in) {
  std::vector<std::pair<char, std::string>> rows;   // sense, name (objective row excluded)
  std::map<std::string, Terms> row_terms;
  std::map<std::string, double> rhs;
  std::string objective_row;
  Terms objective_terms;
  double objective_constant = 0.0;
  std::vector<std::string> variables;               // first-appearance order
  std::map<std::string, Bound> bounds;
  std::string section, line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '*') continue;
    std::istringstream ss(line);
    std::vector<std::string> f;
    for (std::string w; ss >> w;) f.push_back(w);
    if (f.empty()) continue;
    if (line[0] != ' ' && line[0] != '\t') {
      section = f[0];
      if (section == "ENDATA") break;
      continue;
    }
    if (section == "ROWS") {
      if (f[0] == "N") {
        if (objective_row.empty()) objective_row = f[1];
      } else {
        rows.emplace_back(f[0][0], f[1]);
        row_terms[f[1]];
      }
    } else if (section == "COLUMNS") {
      if (f.size() >= 3 && f[1] == "'MARKER'") continue;
      if (!bounds.count(f[0])) {
        bounds[f[0]] = Bound();
        variables.push_back(f[0]);
      }
      for (size_t k = 1; k + 1 < f.size(); k += 2) {
        const double v = std::stod(f[k + 1]);
        if (f[k] == objective_row) objective_terms.emplace_back(f[0], v);
        else if (row_terms.count(f[k])) row_terms[f[k]].emplace_back(f[0], v);
      }
    } else if (section == "RHS") {
      const size_t first = f.size() % 2 == 1 ? 1 : 0;   // optional RHS-set name
      for (size_t k = first; k + 1 < f.size(); k += 2) {
        const double v = std::stod(f[k + 1]);
        if (f[k] == objective_row) objective_constant = -v;
        else rhs[f[k]] = v;
      }
    } else if (section == "BOUNDS") {
      Bound& b = bounds[f[2]];
      const std::string& kind = f[0];
      if (kind == "FR") { b.has_low = b.has_up = false; continue; }
      if (kind == "BV") { b.has_low = b.has_up = true; b.low = 0.0; b.up = 1.0; continue; }
      if (kind == "MI") { b.has_low = false; continue; }
      if (kind == "PL") { b.has_up = false; continue; }
      const double v = std::stod(f[3]);
      if (kind == "LO") { b.has_low = true; b.low = v; }
      else if (kind == "UP") { b.has_up = true; b.up = v; }
      else if (kind == "FX") { b.has_low = b.has_up = true; b.low = b.up = v; }
      else throw std::runtime_error("unsupported MPS bound type " + kind);
    } else if (section == "RANGES") {
      throw std::runtime_error("MPS RANGES sections are not supported");
    }
  }
  std::map<std::string, std::string> rename;
  for (size_t k = 0; k < variables.size(); k++) rename[variables[k]] = "x" + std::to_string(k);
  std::string out = "min " + WriteExpression(objective_terms, objective_constant, rename) + "\nst\n";
  for (const auto& row : rows) {
    const char* rel = row.first == 'E' ? "=" : (row.first == 'G' ? ">=" : "<=");
    auto it = rhs.find(row.second);
    out += WriteExpression(row_terms[row.second], 0.0, rename) + rel + F(it == rhs.end() ? 0.0 : it->second) + "\n";
  }
  std::vector<std::string> by_name = variables;
  std::sort(by_name.begin(), by_name.end());
  for (const auto& var : by_name) {
    const Bound& b = bounds[var];
    if (b.has_low) out += rename[var] + ">=" + F(b.low) + "\n";
    if (b.has_up) out += rename[var] + "<=" + F(b.up) + "\n";
  }
  return out;
}
