// Algebra primitives of the LP front-end: variables, tagged numbers, linear
// expressions, constraints and the objective.  Same type and member names,
// argument meaning and observable behaviour (ordering, zero handling, string
// formats) as the reference's base.h / base.cc, which the text parser and the
// model transforms are written against; the implementation is independent.
#pragma once

#include <sys/time.h>

#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

enum DataType { UNKNOWN, FLOAT, INTEGER };   // reference base.h:24-28

const std::string kUndefined = "undefined";

typedef double real_t;                        // reference base.h:32

// A decision variable: name + numeric kind (reference base.h:34-53).
struct Variable {
  Variable() : variable_name(kUndefined), type(UNKNOWN) {}
  Variable(std::string name, DataType data_type = FLOAT) : variable_name(std::move(name)), type(data_type) {}

  std::string ToString() const { return variable_name; }
  bool IsUndefined() const { return variable_name == kUndefined; }
  void To(DataType target) {
    if (type == target) return;
    if (type == UNKNOWN || target == UNKNOWN) throw std::runtime_error("Cannot convert unknown variable type");
    type = target;
  }

  std::string variable_name;
  DataType type;
};

bool operator==(const Variable& lhs, const Variable& rhs);
bool operator!=(const Variable& lhs, const Variable& rhs);
// Ordering that fixes the tableau's column order (reference base.cc:11-29):
// by kind, then numerically for two "dual<k>" or two "artificial<k>" names,
// otherwise by plain string comparison.
bool operator<(const Variable& lhs, const Variable& rhs);

// Tolerance predicates.  They take float on purpose: callers pass doubles and
// the narrowing happens before the comparison (reference base.h:61-66).
const float kEpsilonF = 1e-6f;
inline bool _IsNegative(float value) { return value <= -kEpsilonF; }
inline bool _IsPositive(float value) { return value >= kEpsilonF; }
inline bool _IsZero(float value) { return std::fabs(value) < kEpsilonF; }
inline bool _IsNonNegative(float value) { return value > -kEpsilonF; }

// Tagged int/double number (reference base.h:68-163).
struct Num {
  Num() : type(UNKNOWN), float_value(0), int_value(0) {}
  Num(real_t val) : type(FLOAT), float_value(val), int_value(0) {}
  Num(int val) : type(INTEGER), float_value(0), int_value(val) {}

  bool IsZero() const { return Dispatch(_IsZero(float_value), int_value == 0); }
  bool IsOne() const { return Dispatch(std::fabs(float_value - 1.0) < kEpsilonF, int_value == 1); }
  bool IsNegative() const { return Dispatch(_IsNegative(float_value), int_value < 0); }
  bool IsNonNegative() const { return Dispatch(_IsNonNegative(float_value), int_value >= 0); }
  bool IsPositive() const { return Dispatch(_IsPositive(float_value), int_value >= 0); }

  std::string ToString() const {
    if (type == FLOAT) return std::to_string(float_value);
    if (type == INTEGER) return std::to_string(int_value);
    throw std::runtime_error("ToString error\n");
  }

  void To(DataType target) {
    if (type == target) return;
    type = target;
    if (target == UNKNOWN) throw std::runtime_error("Cannot convert unknown data type");
    if (target == FLOAT) float_value = int_value;
    else int_value = (int)float_value;
  }

  DataType type;
  real_t float_value;
  int int_value;

 private:
  bool Dispatch(bool if_float, bool if_int) const {
    if (type == FLOAT) return if_float;
    if (type == INTEGER) return if_int;
    throw std::runtime_error("IsZero error\n");
  }
};

const Num kIntZero = Num(0);
const Num kFloatZero = Num(0.0f);
const Num kIntOne = Num(1);
const Num kFloatOne = Num(1.0f);
const Num kFloatMax = Num(std::numeric_limits<real_t>::max());
const Num kFloatMin = Num(std::numeric_limits<real_t>::lowest());
const Num kIntMax = Num(std::numeric_limits<int>::max());
const Num kIntMin = Num(std::numeric_limits<int>::min());
const Num kEpsilon = Num(kEpsilonF);

bool operator==(const Num& lhs, const Num& rhs);
bool operator!=(const Num& lhs, const Num& rhs);
void BinaryOp(Num& lhs, const Num rhs, char op);
void operator*=(Num& lhs, const Num rhs);
Num operator*(const Num lhs, const Num rhs);
void operator+=(Num& lhs, const Num rhs);
Num operator+(const Num lhs, const Num rhs);
void operator/=(Num& lhs, const Num rhs);
Num operator/(const Num lhs, const Num rhs);
void operator-=(Num& lhs, const Num rhs);
Num operator-(const Num lhs, const Num rhs);
bool operator<(const Num lhs, const Num rhs);
bool operator<=(const Num lhs, const Num rhs);
bool operator>(const Num lhs, const Num rhs);
bool operator>=(const Num lhs, const Num rhs);
Num operator-(const Num num);

// sum_i coeff_i * var_i + constant (reference base.h:200-231).
struct Expression {
  Expression(Num num) : constant(num) {}
  Expression(Variable var) : constant(var.type == FLOAT ? kFloatZero : kIntZero) {
    variable_coeff.emplace(var, var.type == FLOAT ? kFloatOne : kIntOne);
  }

  Num GetCoeffOf(Variable var) {
    auto it = variable_coeff.find(var);
    if (it != variable_coeff.end()) return it->second;
    return Num(var.type == FLOAT ? kFloatZero : kIntZero);
  }
  // Exact zeros are never stored.
  void SetCoeffOf(Variable var, Num coeff) {
    variable_coeff.erase(var);
    if (coeff == kIntZero || coeff == kFloatZero) return;
    variable_coeff.emplace(var, coeff);
  }

  std::string ToString() const {
    std::string out;
    for (const auto& term : variable_coeff) out += term.second.ToString() + " * " + term.first.ToString() + " + ";
    return out + constant.ToString();
  }

  std::map<Variable, Num> variable_coeff;
  Num constant;
};

bool operator==(const Expression lhs, const Expression rhs);
bool operator!=(const Expression lhs, const Expression rhs);
Expression operator*(const Variable var, const Num num);
Expression operator*(const Num num, const Variable var);
void operator+=(Expression& lhs, const Expression rhs);
Expression operator+(const Expression exp1, const Expression exp2);
void operator-=(Expression& lhs, const Expression rhs);
Expression operator-(const Expression exp1, const Expression exp2);
void operator*=(Expression& exp, const Num multiplier);
Expression operator*(const Expression exp1, const Num multiplier);
Expression operator*(const Num multiplier, const Expression exp1);
void operator/=(Expression& exp, const Num multiplier);
Expression operator/(const Expression exp1, const Num multiplier);
void ReplaceVariableWithExpression(Expression& expression, Variable var, Expression substitution);

// expression <=/>=/= compare (reference base.h:257-308).
struct Constraint {
  enum Type { LE, GE, EQ };
  Constraint(DataType type) : expression(kFloatZero), data_type(type) {
    if (type == FLOAT) { expression = Expression(kFloatZero); compare = Num(kFloatZero); }
    else if (type == INTEGER) { expression = Expression(kIntZero); compare = Num(kIntZero); }
    else throw std::runtime_error("Unknown constraint data type");
  }
  void SetConstant(Num constant) { expression.constant = constant; }
  void SetCompare(Num value) { compare = value; }
  void SetDataType(DataType type) { data_type = type; }
  void SetEquationType(Type type) { equation_type = type; }

  std::string ToString() {
    static const char* kSym[] = {"<=", ">=", "="};
    return expression.ToString() + " " + kSym[equation_type] + " " + compare.ToString();
  }

  Expression expression;
  Num compare;
  Type equation_type = EQ;
  DataType data_type;
};

bool operator==(const Constraint lhs, const Constraint rhs);

// min/max expression (reference base.h:312-354).
struct OptimizationObject {
  enum Type { MIN, MAX };
  OptimizationObject(DataType type) : expression(kFloatZero), data_type(type) {
    if (type == FLOAT) expression = Expression(kFloatZero);
    else if (type == INTEGER) expression = Expression(kIntZero);
    else throw std::runtime_error("Unknown opt object data type");
  }
  void SetOptType(Type type) { opt_type = type; }
  void SetDataType(DataType type) { data_type = type; }
  std::string ToString() { return std::string(opt_type == MIN ? "min " : "max ") + expression.ToString(); }

  Expression expression;
  Type opt_type = MIN;
  DataType data_type;
};

// Wall-clock stopwatch in microseconds (reference base.h:356-368).
class Timer {
 public:
  Timer() { Reset(); }
  void Reset() { gettimeofday(&start_, nullptr); }
  void Stop() { gettimeofday(&end_, nullptr); }
  long Delta() { return (end_.tv_usec - start_.tv_usec) + (end_.tv_sec - start_.tv_sec) * 1000L * 1000L; }

 private:
  struct timeval start_, end_;
};
