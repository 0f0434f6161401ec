// Out-of-line part of algebra.h.  Behavioural contract: reference base.cc
// (ordering :11-29, Num arithmetic :48-178, Expression algebra :180-286).
#include "algebra.h"

namespace {

bool HasPrefix(const std::string& s, const char* prefix, size_t len) { return s.compare(0, len, prefix) == 0; }

// INTEGER op FLOAT promotes the left operand first (reference base.cc:107-157).
void Promote(Num& lhs, const Num& rhs) {
  if (lhs.type == INTEGER && rhs.type == FLOAT) {
    lhs.type = FLOAT;
    lhs.float_value = lhs.int_value;
  }
}

template <typename T>
T Apply(T a, T b, char op) {
  switch (op) {
    case '*': return a * b;
    case '+': return a + b;
    case '/': return a / b;
    case '-': return a - b;
    default: throw std::runtime_error(std::string("Unknown binary operator ") + op);
  }
}

float AsFloat(const Num& n) { return n.type == FLOAT ? (float)n.float_value : (float)n.int_value; }

}  // namespace

bool operator==(const Variable& lhs, const Variable& rhs) { return lhs.variable_name == rhs.variable_name && lhs.type == rhs.type; }
bool operator!=(const Variable& lhs, const Variable& rhs) { return !(lhs == rhs); }

bool operator<(const Variable& lhs, const Variable& rhs) {
  if (lhs.type != rhs.type) return lhs.type < rhs.type;
  const std::string &a = lhs.variable_name, &b = rhs.variable_name;
  if (HasPrefix(a, "dual", 4) && HasPrefix(b, "dual", 4)) return std::stoi(a.substr(4)) < std::stoi(b.substr(4));
  if (HasPrefix(a, "artificial", 10) && HasPrefix(b, "artificial", 10)) return std::stoi(a.substr(10)) < std::stoi(b.substr(10));
  return a < b;
}

bool operator==(const Num& lhs, const Num& rhs) {
  if (lhs.type != rhs.type) return false;
  if (lhs.type == FLOAT) return lhs.float_value == rhs.float_value;
  if (lhs.type == INTEGER) return lhs.int_value == rhs.int_value;
  throw std::runtime_error("Num == Num error\n");
}
bool operator!=(const Num& lhs, const Num& rhs) { return !(lhs == rhs); }

void BinaryOp(Num& lhs, const Num rhs, char op) {
  if (lhs.type == FLOAT) {
    lhs.float_value = Apply<real_t>(lhs.float_value, rhs.type == FLOAT ? rhs.float_value : (real_t)rhs.int_value, op);
    return;
  }
  if (lhs.type == INTEGER) {
    lhs.int_value = Apply<int>(lhs.int_value, rhs.int_value, op);
    return;
  }
  throw std::runtime_error("Unknown binary element type");
}

void operator*=(Num& lhs, const Num rhs) { Promote(lhs, rhs); BinaryOp(lhs, rhs, '*'); }
void operator+=(Num& lhs, const Num rhs) { Promote(lhs, rhs); BinaryOp(lhs, rhs, '+'); }
void operator/=(Num& lhs, const Num rhs) { Promote(lhs, rhs); BinaryOp(lhs, rhs, '/'); }
void operator-=(Num& lhs, const Num rhs) { Promote(lhs, rhs); BinaryOp(lhs, rhs, '-'); }
Num operator*(const Num lhs, const Num rhs) { Num out = lhs; out *= rhs; return out; }
Num operator+(const Num lhs, const Num rhs) { Num out = lhs; out += rhs; return out; }
Num operator/(const Num lhs, const Num rhs) { Num out = lhs; out /= rhs; return out; }
Num operator-(const Num lhs, const Num rhs) { Num out = lhs; out -= rhs; return out; }

// Ordering compares in float, like the reference (base.cc:159-171).
bool operator<(const Num lhs, const Num rhs) { return AsFloat(lhs) < AsFloat(rhs); }
bool operator<=(const Num lhs, const Num rhs) { return AsFloat(lhs) <= AsFloat(rhs); }
bool operator>(const Num lhs, const Num rhs) { return !(lhs <= rhs); }
bool operator>=(const Num lhs, const Num rhs) { return !(lhs < rhs); }

Num operator-(const Num num) {
  Num out = num;
  out.float_value = -out.float_value;
  out.int_value = -out.int_value;
  return out;
}

bool operator==(const Expression lhs, const Expression rhs) {
  if (lhs.constant != rhs.constant || lhs.variable_coeff.size() != rhs.variable_coeff.size()) return false;
  for (const auto& term : lhs.variable_coeff) {
    auto it = rhs.variable_coeff.find(term.first);
    if (it == rhs.variable_coeff.end() || term.second != it->second) return false;
  }
  return true;
}
bool operator!=(const Expression lhs, const Expression rhs) { return !(lhs == rhs); }

Expression operator*(const Variable var, const Num num) {
  Expression out(num.type == FLOAT ? kFloatZero : kIntZero);
  out.SetCoeffOf(var, num);
  return out;
}
Expression operator*(const Num num, const Variable var) { return var * num; }

namespace {
bool IsStoredZero(const Num& n) { return !(n != kFloatZero && n != kIntZero); }
}  // namespace

void operator+=(Expression& lhs, const Expression rhs) {
  for (const auto& term : rhs.variable_coeff) {
    const Num have = lhs.GetCoeffOf(term.first);
    lhs.SetCoeffOf(term.first, IsStoredZero(have) ? term.second : have + term.second);
  }
  lhs.constant += rhs.constant;
}
void operator-=(Expression& lhs, const Expression rhs) {
  for (const auto& term : rhs.variable_coeff) {
    const Num have = lhs.GetCoeffOf(term.first);
    lhs.SetCoeffOf(term.first, IsStoredZero(have) ? Num(-1.0f) * term.second : have - term.second);
  }
  lhs.constant -= rhs.constant;
}
Expression operator+(const Expression exp1, const Expression exp2) { Expression out = exp1; out += exp2; return out; }
Expression operator-(const Expression exp1, const Expression exp2) { Expression out = exp1; out -= exp2; return out; }

// Scaling drops coefficients that become float-zero (reference base.cc:237-279).
void operator*=(Expression& exp, const Num multiplier) {
  if (multiplier == kIntZero || multiplier == kFloatZero) exp = Expression(multiplier);
  for (auto it = exp.variable_coeff.begin(); it != exp.variable_coeff.end();) {
    it->second *= multiplier;
    it = it->second.IsZero() ? exp.variable_coeff.erase(it) : std::next(it);
  }
  exp.constant *= multiplier;
}
void operator/=(Expression& exp, const Num multiplier) {
  for (auto it = exp.variable_coeff.begin(); it != exp.variable_coeff.end();) {
    it->second /= multiplier;
    it = it->second.IsZero() ? exp.variable_coeff.erase(it) : std::next(it);
  }
  exp.constant /= multiplier;
}
Expression operator*(const Expression exp1, const Num multiplier) { Expression out = exp1; out *= multiplier; return out; }
Expression operator*(const Num multiplier, const Expression exp1) { return exp1 * multiplier; }
Expression operator/(const Expression exp1, const Num multiplier) { Expression out = exp1; out /= multiplier; return out; }

void ReplaceVariableWithExpression(Expression& expression, Variable var, Expression substitution) {
  const Num coeff = expression.GetCoeffOf(var);
  expression.SetCoeffOf(var, var.type == FLOAT ? kFloatZero : kIntZero);
  expression += substitution * coeff;
}

bool operator==(const Constraint lhs, const Constraint rhs) {
  return lhs.compare == rhs.compare && lhs.expression == rhs.expression && lhs.equation_type == rhs.equation_type &&
         lhs.data_type == rhs.data_type;
}
