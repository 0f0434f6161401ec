// LPModel: normal-form transforms on the host, tableau solve on the GPU engine.
// Behavioural contract: reference lp.cc:3-306 (transforms, ToTableau,
// PrintTableau, sanity checks) and lp_simplex_tableau.cc (the three hot
// functions, which here forward to the lpamd C ABI).
#include "lp_model.h"

#include <algorithm>
#include <cstdlib>

#include "host_assert.h"
#include "lpamd.h"

namespace {

bool StartsWith(const std::string& s, const std::string& prefix) { return s.rfind(prefix, 0) == 0; }

void CheckEngine(int rc, const char* what) {
  if (rc != LPAMD_OK) throw std::runtime_error(std::string(what) + ": " + lpamd_last_error());
}

}  // namespace

bool IsUserDefined(Variable var) {
  const std::string& n = var.variable_name;
  return !StartsWith(n, kBase) && !StartsWith(n, kSubstitution) && !StartsWith(n, kDual) && !StartsWith(n, kArtificial);
}

void LPModel::SetOptimizationObject(OptimizationObject obj) {
  LP_ASSERT(obj.expression.constant == kFloatZero);
  model_.opt_obj = obj;
}

void LPModel::SetLogEvery(int iters) {
  LP_ASSERT(iters > 0);
  log_every_iters_ = iters;
}

std::string LPModel::ToString() {
  std::string out = model_.opt_obj.ToString() + "\n";
  for (auto& constraint : model_.constraints) out += constraint.ToString() + "\n";
  return out;
}

Variable LPModel::CreateBaseVariable() { return Variable(kBase + std::to_string(base_variable_count_++)); }
Variable LPModel::CreateSubstitutionVariable() { return Variable(kSubstitution + std::to_string(substitution_variable_count_++)); }
Variable LPModel::CreateDualVariable() { return Variable(kDual + std::to_string(dual_variable_count_++)); }
Variable LPModel::CreateArtificialVariable() { return Variable(kArtificial + std::to_string(artificial_variable_count_++)); }

// "-1 * x <= 0" with nothing else in it (reference lp.cc:298-306).
bool LPModel::IsNonNegativeConstraint(const Constraint& constraint) {
  if (constraint.equation_type != Constraint::LE) return false;
  if (!constraint.compare.IsZero() || !constraint.expression.constant.IsZero()) return false;
  if (constraint.expression.variable_coeff.size() != 1) return false;
  return constraint.expression.variable_coeff.begin()->second == Num(-1.0f);
}

// reference lp.cc:55-129
void LPModel::ToStandardForm() {
  // 1. minimisation -> maximisation of the negated objective
  if (model_.opt_obj.opt_type == OptimizationObject::MIN) {
    opt_reverted_ = true;
    model_.opt_obj.expression *= -1.0f;
    model_.opt_obj.opt_type = OptimizationObject::MAX;
  }
  // 2. lhs = rhs  ->  lhs >= rhs (in place) and lhs <= rhs (appended after all original rows)
  std::vector<Constraint> mirrored;
  for (auto& constraint : model_.constraints) {
    if (constraint.equation_type != Constraint::EQ) continue;
    constraint.equation_type = Constraint::GE;
    Constraint le(FLOAT);
    le.SetCompare(constraint.compare);
    le.SetEquationType(Constraint::LE);
    le.expression = constraint.expression;
    mirrored.push_back(le);
  }
  model_.constraints.insert(model_.constraints.end(), mirrored.begin(), mirrored.end());
  // 3. constants to the right-hand side; >= rows negated into <= rows
  for (auto& constraint : model_.constraints) {
    constraint.compare -= constraint.expression.constant;
    constraint.expression.constant = 0.0f;
    if (constraint.equation_type == Constraint::GE) {
      constraint.expression *= -1.0f;
      constraint.compare *= -1.0f;
      constraint.equation_type = Constraint::LE;
    }
  }
  // 4. variables are collected from the constraints only
  for (auto& constraint : model_.constraints)
    for (auto& term : constraint.expression.variable_coeff) non_base_variables_.insert(term.first);
  // 5. rows that only declare x >= 0 are swallowed; every other variable is free
  std::set<Variable> free_vars = non_base_variables_;
  std::vector<size_t> declared;
  for (size_t i = 0; i < model_.constraints.size(); i++) {
    if (!IsNonNegativeConstraint(model_.constraints[i])) continue;
    Variable var = model_.constraints[i].expression.variable_coeff.begin()->first;
    free_vars.erase(var);
    non_negative_variables_.insert(var);
    declared.push_back(i);
  }
  for (auto it = declared.rbegin(); it != declared.rend(); ++it) model_.constraints.erase(model_.constraints.begin() + *it);
  // 6. a free variable x becomes s1 - s2 with s1, s2 >= 0
  for (const Variable& var : free_vars) {
    non_base_variables_.erase(var);
    Variable s1 = CreateSubstitutionVariable(), s2 = CreateSubstitutionVariable();
    Expression split = 1.0f * s1;
    split -= 1.0f * s2;
    raw_variable_expression_.insert(std::make_pair(var, split));
    for (auto& constraint : model_.constraints) ReplaceVariableWithExpression(constraint.expression, var, split);
    ReplaceVariableWithExpression(model_.opt_obj.expression, var, split);
    for (const Variable& s : {s1, s2}) {
      non_base_variables_.insert(s);
      non_negative_variables_.insert(s);
    }
  }
}

// reference lp.cc:135-145
void LPModel::ToSlackForm() {
  LP_ASSERT(StandardFormSanityCheck(*this) == true);
  for (auto& constraint : model_.constraints) {
    Variable slack = CreateBaseVariable();
    constraint.expression = Expression(constraint.compare) - constraint.expression;
    constraint.compare = kFloatZero;
    constraint.equation_type = Constraint::EQ;
    constraint.expression += -1.0f * slack;
    base_variables_.insert(slack);
  }
}

// reference lp.cc:147-197 without the device part: column index = rank of the
// variable in Variable order, RHS at index N, float-zero coefficients dropped.
HostTableau LPModel::BuildHostTableau() {
  variable_to_index_.clear();
  index_to_variable_.clear();
  for (const auto& var : base_variables_) variable_to_index_[var] = -1;
  for (const auto& var : non_base_variables_) variable_to_index_[var] = -1;
  HostTableau h;
  h.N = (int64_t)variable_to_index_.size();
  h.m = (int64_t)model_.constraints.size();
  h.T.assign((size_t)h.m * (h.N + 2), 0.0);
  h.obj.assign(h.N + 2, 0.0);
  h.is_basic.assign(h.N + 2, 0);
  tableau_index_t next_id = 0;
  for (auto& entry : variable_to_index_) {
    h.is_basic[next_id] = base_variables_.count(entry.first) ? 1 : 0;
    index_to_variable_[next_id] = entry.first;
    entry.second = next_id++;
  }
  constant_index_ = next_id;
  int64_t row = 0;
  for (auto& con : model_.constraints) {
    for (auto& term : con.expression.variable_coeff) {
      auto it = variable_to_index_.find(term.first);
      if (it == variable_to_index_.end() || term.second.IsZero()) continue;
      h.T[(size_t)it->second * h.m + row] = term.second.float_value;
    }
    h.T[(size_t)constant_index_ * h.m + row] = con.expression.constant.float_value;
    row++;
  }
  for (auto& term : model_.opt_obj.expression.variable_coeff) {
    auto it = variable_to_index_.find(term.first);
    if (it == variable_to_index_.end() || term.second.IsZero()) continue;
    h.obj[it->second] = term.second.float_value;
  }
  if (!model_.opt_obj.expression.constant.IsZero()) h.obj[constant_index_] = model_.opt_obj.expression.constant.float_value;
  return h;
}

void LPModel::ToTableau(TableauStorageFormat format) {
  if (HasTableau() || rev_engine_ != nullptr) return;
  HostTableau h = BuildHostTableau();
  LP_ASSERT(h.m > 0 && h.N > 0);
  if (format == COLUMN_ONLY) {
    // The storage format upstream picks for the revised simplex (reference solver.cc:139,
    // lp_revised_simplex_tableau_test.cc:14); here it selects the engine that keeps the
    // tableau untouched and carries the basis inverse.
    lpamd_rev_engine* rev = nullptr;
    CheckEngine(lpamd_rev_create(&rev, h.m, h.N, nullptr), "ToTableau: creating the GPU engine failed");
    rev_engine_.reset(rev, [](lpamd_rev_engine* p) { lpamd_rev_destroy(p); });
    CheckEngine(lpamd_rev_upload(rev, h.T.data(), h.m, h.obj.data(), h.is_basic.data(), opt_reverted_ ? 1 : 0),
                "ToTableau: upload failed");
    return;
  }
  int shards = n_gpus_;
  if (shards <= 0) {
    const char* env = std::getenv("LPAMD_GPUS");
    shards = env ? std::atoi(env) : 1;
  }
  shards = (int)std::min<int64_t>(std::max(shards, 1), h.N + 1);   // never more shards than columns
  if (shards > 1) {
    // column shards, one per GPU (reference has no counterpart; results are those of one GPU bit for bit)
    std::vector<int32_t> devices(shards, 0);
    for (int k = 0; k < shards && !share_one_gpu_; k++) devices[k] = k;
    lpamd_group* group = nullptr;
    CheckEngine(lpamd_group_create(&group, h.m, h.N, shards, devices.data(), share_one_gpu_ ? LPAMD_GROUP_LOCAL_EXCHANGE : 0),
                "ToTableau: creating the GPU shards failed");
    group_.reset(group, [](lpamd_group* p) { lpamd_group_destroy(p); });
    CheckEngine(lpamd_group_upload(group, h.T.data(), h.m, h.obj.data(), h.is_basic.data()), "ToTableau: upload failed");
    return;
  }
  lpamd_engine* eng = nullptr;
  CheckEngine(lpamd_create(&eng, h.m, h.N, 0, 1, nullptr), "ToTableau: creating the GPU engine failed");
  engine_.reset(eng, [](lpamd_engine* p) { lpamd_destroy(p); });
  CheckEngine(lpamd_upload(eng, h.T.data(), h.m, h.obj.data(), h.is_basic.data()), "ToTableau: upload failed");
}

// reference lp.cc:199-219: exact zeros are skipped, the RHS entry carries no name.
std::string LPModel::PrintTableau() {
  LP_ASSERT(HasTableau());
  const int64_t m = (int64_t)model_.constraints.size(), W = constant_index_ + 2;
  std::vector<double> T((size_t)m * W), obj(W);
  CheckEngine(group_ ? lpamd_group_download(group_.get(), T.data(), m, obj.data(), nullptr)
                     : lpamd_download(engine_.get(), T.data(), m, obj.data(), nullptr),
              "PrintTableau: download failed");
  auto name_of = [&](int64_t j) {
    auto it = index_to_variable_.find(j);
    return it != index_to_variable_.end() ? it->second.ToString() : Variable(kArtificial + std::to_string(std::max(0, artificial_variable_count_ - 1))).ToString();
  };
  auto print = [&](auto at) {
    std::string line;
    int printed = 0;
    for (int64_t j = 0; j < W; j++) {
      const double v = at(j);
      if (v == 0.0) continue;
      if (printed++ > 0) line += " + ";
      line += std::to_string(v);
      if (j != constant_index_) line += " * " + name_of(j);
    }
    return line;
  };
  std::string out = print([&](int64_t j) { return obj[j]; }) + "\n";
  for (int64_t i = 0; i < m; i++) out += print([&](int64_t j) { return T[(size_t)j * m + i]; }) + "\n";
  return out;
}

void LPModel::SyncBasisFromEngine() {
  std::vector<uint8_t> basic(constant_index_ + 2, 0);
  CheckEngine(group_ ? lpamd_group_download(group_.get(), nullptr, 0, nullptr, basic.data())
                     : lpamd_download(engine_.get(), nullptr, 0, nullptr, basic.data()),
              "reading the basis flags failed");
  base_variables_.clear();
  non_base_variables_.clear();
  for (const auto& entry : index_to_variable_) (basic[entry.first] ? base_variables_ : non_base_variables_).insert(entry.second);
}

// reference lp_simplex_tableau.cc:10-52
void LPModel::TableauPivot(Variable base, Variable non_base, tableau_index_t constraint_id) {
  LP_ASSERT(HasTableau());
  LP_ASSERT(non_base_variables_.find(non_base) != non_base_variables_.end());
  LP_ASSERT(base_variables_.find(base) != base_variables_.end());
  base_variables_.erase(base);
  base_variables_.insert(non_base);
  non_base_variables_.erase(non_base);
  non_base_variables_.insert(base);
  const tableau_index_t d = variable_to_index_[base], e = variable_to_index_[non_base];
  LP_ASSERT(d != constant_index_);
  LP_ASSERT(e != constant_index_);
  LP_ASSERT(constraint_id >= 0);
  CheckEngine(group_ ? lpamd_group_pivot(group_.get(), d, e, constraint_id) : lpamd_pivot(engine_.get(), d, e, constraint_id),
              "TableauPivot failed");
}

// reference lp_simplex_tableau.cc:64-171
Result LPModel::TableauSimplexInitialize() {
  LP_ASSERT(HasTableau());
  int32_t need = 0;
  CheckEngine(group_ ? lpamd_group_need_init(group_.get(), &need) : lpamd_need_init(engine_.get(), &need), "TableauSimplexInitialize");
  if (!need) return SOLVED;
  CreateArtificialVariable();   // the name is consumed exactly as upstream; the column itself lives at index N + 1
  int32_t result = LPAMD_ERROR;
  CheckEngine(group_ ? lpamd_group_phase1(group_.get(), (int32_t)strategy_, &result) : lpamd_phase1(engine_.get(), (int32_t)strategy_, &result),
              "TableauSimplexInitialize");
  SyncBasisFromEngine();
  if (result == LPAMD_NOSOLUTION) return NOSOLUTION;
  LP_ASSERT(result == LPAMD_SOLVED);   // upstream's live asserts (:93, :108, :111, :133, :159) abort here as well
  return SOLVED;
}

// reference lp_simplex_tableau.cc:173-379
Result LPModel::TableauSimplexSolve() {
  LP_ASSERT(HasTableau());
  int32_t need = 0;
  CheckEngine(group_ ? lpamd_group_need_init(group_.get(), &need) : lpamd_need_init(engine_.get(), &need), "TableauSimplexSolve");
  if (need) {
    Result r = TableauSimplexInitialize();
    if (r == NOSOLUTION) return NOSOLUTION;
    LP_ASSERT(r == SOLVED);
  }
  // Only MAX_COST and MAX_REDUCTION have a pricing rule on this path; upstream aborts on
  // its first live assert in TableauPivot for any other strategy.
  LP_ASSERT(strategy_ == MAX_COST || strategy_ == MAX_REDUCTION);
  int32_t result = LPAMD_ERROR;
  int64_t pivots = 0;
  CheckEngine(group_ ? lpamd_group_solve(group_.get(), (int32_t)strategy_, -1, &result, &pivots)
                     : lpamd_solve(engine_.get(), (int32_t)strategy_, -1, &result, &pivots),
              "TableauSimplexSolve");
  SyncBasisFromEngine();
  if (result == LPAMD_SOLVED) {
    simplex_solution_ = GetTableauSimplexSolution();
    simplex_optimum_ = GetTableauSimplexOptimum();
    return SOLVED;
  }
  if (result == LPAMD_UNBOUNDED) return UNBOUNDED;
  if (result == LPAMD_NOSOLUTION) return NOSOLUTION;
  return ERROR;
}

// reference lp_simplex_tableau.cc:381-410
std::map<Variable, Num> LPModel::GetTableauSimplexSolution() {
  LP_ASSERT(HasTableau());
  std::vector<double> x(constant_index_ + 2, 0.0);
  CheckEngine(group_ ? lpamd_group_column_values(group_.get(), x.data()) : lpamd_column_values(engine_.get(), x.data()),
              "GetTableauSimplexSolution");
  std::map<Variable, Num> all_sol, sol;
  for (const auto& var : non_base_variables_) {
    all_sol[var] = 0.0f;
    if (IsUserDefined(var) || IsOverriddenAsUserDefined(var)) sol[var] = 0.0f;
  }
  for (const auto& var : base_variables_) {
    all_sol[var] = Num(x[variable_to_index_[var]]);
    if (IsUserDefined(var) || IsOverriddenAsUserDefined(var)) sol[var] = all_sol[var];
  }
  for (const auto& entry : raw_variable_expression_) {
    Expression exp = entry.second;
    while (!exp.variable_coeff.empty()) {
      const Variable head = exp.variable_coeff.begin()->first;
      ReplaceVariableWithExpression(exp, head, all_sol[head]);
    }
    sol[entry.first] = exp.constant;
  }
  return sol;
}

// reference lp_simplex_tableau.cc:412-414
Num LPModel::GetTableauSimplexOptimum() {
  LP_ASSERT(HasTableau());
  double c = 0.0;
  CheckEngine(group_ ? lpamd_group_objective_constant(group_.get(), &c) : lpamd_objective_constant(engine_.get(), &c),
              "GetTableauSimplexOptimum");
  return (opt_reverted_ ? -1 : 1) * c;
}

// reference lp_revised_simplex_tableau.cc:92-197.  The feasibility phase is part of the
// engine's solve; upstream only reaches this method from TableauRevisedSimplexSolve.
Result LPModel::TableauRevisedSimplexInitialize() {
  LP_ASSERT(rev_engine_ != nullptr);
  return SOLVED;
}

// reference lp_revised_simplex_tableau.cc:199-302
Result LPModel::TableauRevisedSimplexSolve() {
  LP_ASSERT(rev_engine_ != nullptr);
  LP_ASSERT(model_.opt_obj.opt_type == OptimizationObject::MAX);
  int32_t result = LPAMD_ERROR;
  int64_t pivots = 0;
  CheckEngine(lpamd_rev_solve(rev_engine_.get(), -1, &result, &pivots), "TableauRevisedSimplexSolve");
  if (result == LPAMD_UNBOUNDED) return UNBOUNDED;
  if (result == LPAMD_NOSOLUTION) return NOSOLUTION;
  if (result != LPAMD_SOLVED) return ERROR;
  double optimum = 0.0;
  CheckEngine(lpamd_rev_optimum(rev_engine_.get(), &optimum), "TableauRevisedSimplexSolve");
  revised_simplex_optimum_ = Num(optimum);
  // :259-277: basic_feasible_solution scattered over the variables, then the
  // substituted raw variables are rebuilt from their expressions.
  std::vector<double> x(constant_index_ + 2, 0.0);
  CheckEngine(lpamd_rev_column_values(rev_engine_.get(), x.data()), "TableauRevisedSimplexSolve");
  std::map<Variable, Num> all_sol;
  revised_simplex_solution_.clear();
  for (const auto& entry : variable_to_index_) all_sol[entry.first] = Num(x[entry.second]);
  for (const auto& entry : variable_to_index_)
    if (IsUserDefined(entry.first) || IsOverriddenAsUserDefined(entry.first)) revised_simplex_solution_[entry.first] = all_sol[entry.first];
  for (const auto& entry : raw_variable_expression_) {
    Expression exp = entry.second;
    while (!exp.variable_coeff.empty()) {
      const Variable head = exp.variable_coeff.begin()->first;
      ReplaceVariableWithExpression(exp, head, all_sol[head]);
    }
    revised_simplex_solution_[entry.first] = exp.constant;
  }
  return SOLVED;
}

// reference lp_revised_simplex_tableau.cc:304-310
Num LPModel::GetTableauRevisedSimplexOptimum() { return revised_simplex_optimum_; }
std::map<Variable, Num> LPModel::GetTableauRevisedSimplexSolution() { return revised_simplex_solution_; }

int64_t LPModel::TableauPivotCount() {
  int64_t n = 0;
  if (rev_engine_ != nullptr) {
    CheckEngine(lpamd_rev_pivot_count(rev_engine_.get(), &n), "TableauPivotCount");
    return n;
  }
  CheckEngine(group_ ? lpamd_group_pivot_count(group_.get(), &n) : lpamd_pivot_count(engine_.get(), &n), "TableauPivotCount");
  return n;
}

std::vector<std::pair<int64_t, int64_t>> LPModel::TableauPivotTrace() {
  const int64_t n = TableauPivotCount();
  std::vector<int64_t> e(std::max<int64_t>(n, 1)), r(std::max<int64_t>(n, 1));
  int64_t got = 0;
  if (rev_engine_ != nullptr) CheckEngine(lpamd_rev_trace(rev_engine_.get(), e.data(), r.data(), n, &got), "TableauPivotTrace");
  else if (group_ != nullptr) CheckEngine(lpamd_group_trace(group_.get(), e.data(), r.data(), n, &got), "TableauPivotTrace");
  else CheckEngine(lpamd_trace(engine_.get(), e.data(), r.data(), n, &got), "TableauPivotTrace");
  std::vector<std::pair<int64_t, int64_t>> out;
  for (int64_t i = 0; i < n; i++) out.emplace_back(e[i], r[i]);
  return out;
}

// reference lp.cc:264-296
bool StandardFormSanityCheck(LPModel model) {
  if (model.model_.opt_obj.opt_type != OptimizationObject::MAX) return false;
  for (auto& constraint : model.model_.constraints)
    if (constraint.equation_type != Constraint::LE) return false;
  for (const auto& var : model.non_base_variables_)
    if (!model.non_negative_variables_.count(var)) return false;
  return true;
}

bool SlackFormSanityCheck(LPModel model) {
  std::set<Variable> seen;
  for (auto& constraint : model.model_.constraints) {
    if (constraint.equation_type != Constraint::EQ) return false;
    int slack_terms = 0;
    for (auto& term : constraint.expression.variable_coeff) {
      if (!model.base_variables_.count(term.first)) continue;
      if (!seen.insert(term.first).second) return false;
      if (term.second != Num(-1.0f)) return false;
      slack_terms++;
    }
    if (slack_terms != 1) return false;
  }
  return true;
}
