// LPModel: the LP in algebraic form, its normal-form transforms, and the dense
// simplex-tableau solve running on the GPU engine (C ABI in include/lpamd.h).
//
// Drop-in surface for the tableau path of the reference's lp.h: same class,
// enum and method names (ToStandardForm, ToSlackForm, ToTableau, PrintTableau,
// SetPivotingStrategy, TableauPivot, TableauSimplexInitialize,
// TableauSimplexSolve, GetTableauSimplexOptimum, GetTableauSimplexSolution,
// ToString, ...).  The reference keeps `Tableau<real_t>* tableau_` /
// `List<real_t>* opt_obj_tableau_` from a header-only CPU library
// (reference lp.h:273-274); here the tableau lives in HBM behind an
// lpamd_engine handle and the three hot functions forward to it.
// The reference's other solvers (map simplex, revised, dual, interior point,
// column generation, ILP) are not part of this path and are not provided.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "algebra.h"

struct lpamd_engine;
struct lpamd_rev_engine;
struct lpamd_group;

typedef int64_t tableau_index_t;
typedef int64_t tableau_size_t;
// Accepted for source compatibility; the device layout is always column-major.
enum TableauStorageFormat { ROW_ONLY, COLUMN_ONLY, ROW_AND_COLUMN };

const std::string kBase = "base";
const std::string kSubstitution = "subst";
const std::string kDual = "dual";
const std::string kArtificial = "artificial";

bool IsUserDefined(Variable var);

struct Model {
  std::vector<Constraint> constraints;
  OptimizationObject opt_obj;
};

enum Result { ERROR, UNBOUNDED, NOSOLUTION, SOLVED };

// Dense host image of the tableau right after ToTableau: what gets uploaded.
struct HostTableau {
  int64_t m = 0, N = 0;             // rows, variables; columns = N + 2 (RHS at N, artificial at N + 1)
  std::vector<double> T;            // column-major, leading dimension m
  std::vector<double> obj;          // N + 2
  std::vector<uint8_t> is_basic;    // N + 2
};

class LPModel {
 public:
  LPModel(Model model) : model_(model) {}
  LPModel() : model_({{}, OptimizationObject(FLOAT)}) {}
  // Copies share the device tableau, as copies of the reference's LPModel share its raw tableau pointers
  // (reference lp.h:273-274); unlike there, the last copy to go frees it.
  LPModel(const LPModel&) = default;
  LPModel& operator=(const LPModel&) = default;

  void AddConstraint(Constraint constraint) { model_.constraints.push_back(constraint); }
  void SetOptimizationObject(OptimizationObject obj);

  void SetEnableLogging(bool enable_logging) { enable_logging_ = enable_logging; }
  void SetLogEvery(int iters);

  // max objective, every constraint `sum c_i x_i <= b`, every variable >= 0.
  void ToStandardForm();
  // One slack ("base") variable per constraint: 0 = b_i - sum a_ij x_j - s_i.
  void ToSlackForm();
  // Builds the dense tableau and moves it into HBM (needs a GPU).
  void ToTableau(TableauStorageFormat format = ROW_AND_COLUMN);
  // Column shards for the next ToTableau(): n_gpus > 1 deals the columns over GPUs 0..n_gpus-1 (RCCL
  // inside the engine).  share_one_gpu keeps all shards on GPU 0 with an in-process exchange (a one-GPU
  // machine can walk the multi-GPU code this way).  Without a call the environment variable LPAMD_GPUS
  // decides, default 1.  The revised simplex runs on one GPU.
  void SetGpus(int n_gpus, bool share_one_gpu = false) { n_gpus_ = n_gpus; share_one_gpu_ = share_one_gpu; }
  // The dense matrix ToTableau uploads, computed on the host only.
  HostTableau BuildHostTableau();

  std::string PrintTableau();

  enum PivotingStrategy { PIVOTING_UNKNOWN, SMALLEST_SUBSCRIPT, MAX_COST, MAX_REDUCTION };
  void SetPivotingStrategy(PivotingStrategy strategy) { strategy_ = strategy; }

  void TableauPivot(Variable base, Variable non_base, tableau_index_t constraint_id);
  Result TableauSimplexInitialize();
  Result TableauSimplexSolve();
  Num GetTableauSimplexOptimum();
  std::map<Variable, Num> GetTableauSimplexSolution();

  // Revised simplex on the tableau; the model must have gone through
  // ToTableau(COLUMN_ONLY), the format upstream's solver selects for it.
  Result TableauRevisedSimplexInitialize();
  Result TableauRevisedSimplexSolve();
  Num GetTableauRevisedSimplexOptimum();
  std::map<Variable, Num> GetTableauRevisedSimplexSolution();

  virtual bool IsBaseVariable(Variable var) { return base_variables_.count(var) != 0; }
  std::set<Variable> GetBaseVariables() { return base_variables_; }
  virtual void AddBaseVariable(Variable var) { base_variables_.insert(var); }
  virtual ~LPModel() = default;

  friend bool StandardFormSanityCheck(LPModel model);
  friend bool SlackFormSanityCheck(LPModel model);

  std::map<Variable, Expression> GetRawVariableExpression() { return raw_variable_expression_; }

  std::string ToString();

  Variable CreateBaseVariable();
  Variable CreateSubstitutionVariable();
  Variable CreateDualVariable();
  Variable CreateArtificialVariable();

  // Number of pivots the engine has applied, and their (entering column, row) pairs.
  int64_t TableauPivotCount();
  std::vector<std::pair<int64_t, int64_t>> TableauPivotTrace();

 private:
  bool IsNonNegativeConstraint(const Constraint& constraint);
  void OverrideAsUserDefined(Variable var) { overrided_as_user_defined_variables_.insert(var); }
  bool IsOverriddenAsUserDefined(Variable var) { return overrided_as_user_defined_variables_.count(var) != 0; }
  void SyncBasisFromEngine();

  Model model_;
  std::set<Variable> base_variables_;
  std::set<Variable> non_base_variables_;
  bool opt_reverted_ = false;
  std::set<Variable> non_negative_variables_;
  std::map<Variable, Expression> raw_variable_expression_;

  // Device state, shared by copies of the model and released with the last of them.
  std::shared_ptr<lpamd_rev_engine> rev_engine_;   // revised simplex: tableau + basis inverse on one GPU
  std::shared_ptr<lpamd_engine> engine_;           // the tableau on one GPU
  std::shared_ptr<lpamd_group> group_;             // the tableau as column shards over several GPUs
  int n_gpus_ = 0;                                 // 0: LPAMD_GPUS or 1
  bool share_one_gpu_ = false;
  bool HasTableau() const { return engine_ != nullptr || group_ != nullptr; }
  std::map<Variable, tableau_index_t> variable_to_index_;
  std::map<tableau_index_t, Variable> index_to_variable_;
  tableau_index_t constant_index_ = -1;

  std::set<Variable> overrided_as_user_defined_variables_;
  Num simplex_optimum_;
  std::map<Variable, Num> simplex_solution_;
  Num revised_simplex_optimum_;
  std::map<Variable, Num> revised_simplex_solution_;

  int base_variable_count_ = 0;
  int substitution_variable_count_ = 0;
  int dual_variable_count_ = 0;
  int artificial_variable_count_ = 0;

  bool enable_logging_ = false;
  int log_every_iters_ = 1;

  PivotingStrategy strategy_ = MAX_COST;
};

bool StandardFormSanityCheck(LPModel model);
bool SlackFormSanityCheck(LPModel model);
