// solver <input-file> [solver-algo] [pivot-strategy] [--gpus N | --shards K] [--time] [--trace]
// Command line of the reference's solver.cc for the tableau path: parses the LP
// text file, runs `simplex_tableau` or `revised_simplex_tableau` on the GPU engine
// and prints the optimum followed by `name = value` lines (reference
// solver.cc:123-171).  The other algorithms belong to the reference's CPU-only
// solvers and are rejected.  A file ending in .mps is first converted the way the reference's
// tools/mps_load.py does, and a file in netlib's compressed form (datasets/netlib/<name> as shipped) is
// expanded to MPS before that, as tools/emps.c does.  --gpus N deals the tableau columns over N GPUs (RCCL inside the engine;
// LPAMD_GPUS=N does the same without touching the command line); --shards K keeps K column shards on
// GPU 0 with an in-process exchange, which walks the same multi-GPU code on a one-GPU machine.
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "lp_model.h"
#include "lp_text.h"
#include "mps_text.h"
#include "netlib_text.h"

namespace {

std::string Lower(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  return s;
}

LPModel::PivotingStrategy ParsePivotingStrategy(const std::string& text) {
  const std::string t = Lower(text);
  if (t == "smallest_subscript") return LPModel::SMALLEST_SUBSCRIPT;
  if (t == "max_cost") return LPModel::MAX_COST;
  if (t == "max_reduction") return LPModel::MAX_REDUCTION;
  return LPModel::PIVOTING_UNKNOWN;
}

bool EndsWith(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

}  // namespace

static int Run(int argc, char** argv);

int main(int argc, char** argv) {
  try {
    return Run(argc, argv);
  } catch (const std::exception& ex) {   // parser errors and engine failures (reference: uncaught std::runtime_error)
    std::cerr << "solver: " << ex.what() << "\n";
    return 3;
  }
}

static int Run(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "usage: " << argv[0] << " input-file [solver-algo] [pivot-strategy]\n";
    return -1;
  }
  const std::string path = argv[1];
  const bool has_algo = argc >= 3 && argv[2][0] != '-';
  const std::string algo = has_algo ? Lower(argv[2]) : "simplex_tableau";
  LPModel::PivotingStrategy strategy = LPModel::MAX_COST;
  if (has_algo && argc >= 4 && argv[3][0] != '-') strategy = ParsePivotingStrategy(argv[3]);
  bool print_trace = false;
  bool print_time = false;   // --time: wall time of the solve call and its pivot count on stderr (stdout stays the reference's)
  int n_gpus = 0, n_shards = 0;
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    print_trace |= a == "--trace";
    print_time |= a == "--time";
    if (a == "--gpus" && i + 1 < argc) n_gpus = std::atoi(argv[++i]);
    else if (a == "--shards" && i + 1 < argc) n_shards = std::atoi(argv[++i]);
  }

  Parser parser;
  Model model = {{}, OptimizationObject(FLOAT)};
  std::ifstream file(path);
  if (EndsWith(Lower(path), ".mps") || (file.is_open() && LooksLikeCompressedNetlib(file))) {
    model = parser.Parse(MpsToLpText(path));
  } else {
    model = parser.Parse(file);
  }
  // An LP whose constraints mention no FLOAT variable is an integer program
  // (reference solver.cc:79-99); that path is not part of this engine.
  bool any_float = false;
  for (auto& con : model.constraints)
    for (auto& term : con.expression.variable_coeff) any_float |= term.first.type == FLOAT;
  const bool revised = algo == "revised_simplex_tableau";
  if (!any_float || (algo != "simplex_tableau" && !revised)) {
    std::cerr << "only `simplex_tableau` and `revised_simplex_tableau` on continuous LPs run on the GPU engine (got algo `" << algo << "`)\n";
    std::cout << "Error\n";
    return 2;
  }

  LPModel lp_model(model);
  lp_model.SetPivotingStrategy(strategy);
  if (n_shards > 1) lp_model.SetGpus(n_shards, true);
  else if (n_gpus > 0) lp_model.SetGpus(n_gpus);
  lp_model.ToStandardForm();
  lp_model.ToSlackForm();
  Result result = ERROR;
  Num optimum;
  std::map<Variable, Num> solution;
  std::chrono::steady_clock::time_point t0, t1;
  if (revised) {   // reference solver.cc:135-145
    lp_model.ToTableau(COLUMN_ONLY);
    t0 = std::chrono::steady_clock::now();
    result = lp_model.TableauRevisedSimplexSolve();
    t1 = std::chrono::steady_clock::now();
    if (result == SOLVED) {
      optimum = lp_model.GetTableauRevisedSimplexOptimum();
      solution = lp_model.GetTableauRevisedSimplexSolution();
    }
  } else {         // reference solver.cc:123-133
    lp_model.ToTableau();
    t0 = std::chrono::steady_clock::now();
    result = lp_model.TableauSimplexSolve();
    t1 = std::chrono::steady_clock::now();
    if (result == SOLVED) {
      optimum = lp_model.GetTableauSimplexOptimum();
      solution = lp_model.GetTableauSimplexSolution();
    }
  }
  if (result == SOLVED) {
    std::cout << optimum.ToString() << "\n";
    for (auto& entry : solution) std::cout << entry.first.ToString() << " = " << entry.second.ToString() << "\n";
  } else if (result == NOSOLUTION) {
    std::cout << "No Solution\n";
  } else if (result == UNBOUNDED) {
    std::cout << "Unbounded\n";
  } else {
    std::cout << "Error\n";
  }
  if (print_time)
    std::cerr << "solve_ms " << std::chrono::duration<double, std::milli>(t1 - t0).count() << " pivots "
              << lp_model.TableauPivotTrace().size() << "\n";
  if (print_trace)
    for (auto& p : lp_model.TableauPivotTrace()) std::cerr << "pivot " << p.first << " " << p.second << "\n";
  return 0;
}
