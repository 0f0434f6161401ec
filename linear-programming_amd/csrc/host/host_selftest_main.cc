// lp_host_selftest [--gpu] <dir-with-lp-fixtures>
// Known-answer checks of the host layer against the values the reference's own
// unit tests expect (reference parser_test.cc:6-138, lp_test.cc:8-82 for the
// CPU part; lp_test.cc:134-148, 231-267, 472-513 for the tableau path, which
// needs a GPU and runs only with --gpu).
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "lp_model.h"
#include "lp_text.h"

static int g_failed = 0, g_checked = 0;

#define CHECK_EQ(got, want)                                                                       \
  do {                                                                                            \
    g_checked++;                                                                                  \
    auto g_ = (got);                                                                              \
    auto w_ = (want);                                                                             \
    if (!(g_ == w_)) {                                                                            \
      g_failed++;                                                                                 \
      std::cerr << __FILE__ << ":" << __LINE__ << ": mismatch\n  got : " << g_ << "\n  want: " << w_ << "\n"; \
    }                                                                                             \
  } while (0)

static const char* kTypeName[] = {"UNKNOWN", "VAR", "NUM", "MAX", "MIN", "ST", "EQ", "GE", "LE", "ADD", "SUB", "MUL", "EOL", "COMMA"};

// "TYPE:lexeme:start:end" joined by blanks
static std::string Show(const std::vector<Token>& toks) {
  std::ostringstream os;
  for (size_t i = 0; i < toks.size(); i++)
    os << (i ? " " : "") << kTypeName[toks[i].type] << ":" << toks[i].lexim << ":" << toks[i].start_index << ":" << toks[i].end_index;
  return os.str();
}

static LPModel Load(const std::string& dir, const char* name) {
  std::ifstream file(dir + "/" + name);
  Parser parser;
  return LPModel(parser.Parse(file));
}

static void HostCases(const std::string& dir) {
  Lexer lex;
  CHECK_EQ(Show(lex.Scan("max 3 * x + 4 * y")), std::string("MAX:max:0:3 NUM:3:4:5 MUL:*:6:7 VAR:x:8:9 ADD:+:10:11 NUM:4:12:13 MUL:*:14:15 VAR:y:16:17"));
  CHECK_EQ(Show(lex.Scan("st")), std::string("ST:st:0:2"));
  CHECK_EQ(Show(lex.Scan("3 * x + 4 * y -5<=60")),
           std::string("NUM:3:0:1 MUL:*:2:3 VAR:x:4:5 ADD:+:6:7 NUM:4:8:9 MUL:*:10:11 VAR:y:12:13 SUB:-:14:15 NUM:5:15:16 LE:<=:16:18 NUM:60:18:20"));
  CHECK_EQ(Show(lex.Scan("-2.0 * x1 + 3.0 * x2")),
           std::string("SUB:-:0:1 NUM:2.0:1:4 MUL:*:5:6 VAR:x1:7:9 ADD:+:10:11 NUM:3.0:12:15 MUL:*:16:17 VAR:x2:18:20"));

  Parser parser;
  CHECK_EQ(parser.ParseOptimizationObject(lex.Scan("max 3 * x + 4 * y")).ToString(), std::string("max 3.000000 * x + 4.000000 * y + 0.000000"));
  CHECK_EQ(parser.ParseConstraint(lex.Scan("3 * x + 4 * y -5<=60")).ToString(), std::string("3.000000 * x + 4.000000 * y + -65.000000 <= 0.000000"));
  CHECK_EQ(parser.ParseExpression(lex.Scan("3 * x + 4 * y -5")).ToString(), std::string("3.000000 * x + 4.000000 * y + -5.000000"));
  CHECK_EQ(parser.ParseExpression(lex.Scan("-2.0 * x1 + 3.0 * x2")).ToString(), std::string("-2.000000 * x1 + 3.000000 * x2 + 0.000000"));
  {
    LPModel model(parser.Parse(std::string("max 3 * x + 4 * y\nst\n3 * x + 4 * y -5<=60")));
    CHECK_EQ(model.ToString(), std::string("max 3.000000 * x + 4.000000 * y + 0.000000\n3.000000 * x + 4.000000 * y + -65.000000 <= 0.000000\n"));
  }
  CHECK_EQ(Load(dir, "test1.txt").ToString(),
           std::string("max 1.000000 * x1 + 14.000000 * x2 + 6.000000 * x3 + 0.000000\n"
                       "1.000000 * x1 + 1.000000 * x2 + 1.000000 * x3 + -3.000000 <= 0.000000\n"
                       "1.000000 * x1 + -2.000000 <= 0.000000\n"
                       "1.000000 * x3 + -3.000000 <= 0.000000\n"
                       "3.000000 * x2 + 1.000000 * x3 + -6.000000 <= 0.000000\n"
                       "1.000000 * x1 + 0.000000 >= 0.000000\n"
                       "1.000000 * x2 + 0.000000 >= 0.000000\n"
                       "1.000000 * x3 + 0.000000 >= 0.000000\n"));
  {  // free variable x2 is split, the equality becomes two inequalities
    LPModel model = Load(dir, "test0.txt");
    model.ToStandardForm();
    CHECK_EQ(model.ToString(),
             std::string("max -3.000000 * subst0 + 3.000000 * subst1 + 2.000000 * x1 + -0.000000\n"
                         "-1.000000 * subst0 + 1.000000 * subst1 + -1.000000 * x1 + -0.000000 <= -7.000000\n"
                         "-2.000000 * subst0 + 2.000000 * subst1 + 1.000000 * x1 + 0.000000 <= 4.000000\n"
                         "1.000000 * subst0 + -1.000000 * subst1 + 1.000000 * x1 + 0.000000 <= 7.000000\n"));
    CHECK_EQ(StandardFormSanityCheck(model), true);
    Expression split = Variable("subst0") - Variable("subst1");
    auto raw = model.GetRawVariableExpression();
    CHECK_EQ(raw.size(), (size_t)1);
    CHECK_EQ(raw.begin()->first == Variable("x2") && raw.begin()->second == split, true);
    model.ToSlackForm();
    CHECK_EQ(model.ToString(),
             std::string("max -3.000000 * subst0 + 3.000000 * subst1 + 2.000000 * x1 + -0.000000\n"
                         "-1.000000 * base0 + 1.000000 * subst0 + -1.000000 * subst1 + 1.000000 * x1 + -7.000000 = 0.000000\n"
                         "-1.000000 * base1 + 2.000000 * subst0 + -2.000000 * subst1 + -1.000000 * x1 + 4.000000 = 0.000000\n"
                         "-1.000000 * base2 + -1.000000 * subst0 + 1.000000 * subst1 + -1.000000 * x1 + 7.000000 = 0.000000\n"));
    CHECK_EQ(SlackFormSanityCheck(model), true);
  }
  {
    LPModel model = Load(dir, "test3.txt");
    CHECK_EQ(model.ToString(),
             std::string("min -1.000000 * x1 + -1.000000 * x2 + 0.000000\n"
                         "2.000000 * x1 + 1.000000 * x2 + -12.000000 <= 0.000000\n"
                         "1.000000 * x1 + 2.000000 * x2 + -9.000000 <= 0.000000\n"
                         "1.000000 * x1 + 0.000000 >= 0.000000\n"
                         "1.000000 * x2 + 0.000000 >= 0.000000\n"));
    model.ToStandardForm();
    CHECK_EQ(model.ToString(),
             std::string("max 1.000000 * x1 + 1.000000 * x2 + -0.000000\n"
                         "2.000000 * x1 + 1.000000 * x2 + 0.000000 <= 12.000000\n"
                         "1.000000 * x1 + 2.000000 * x2 + 0.000000 <= 9.000000\n"));
    model.ToSlackForm();
    CHECK_EQ(model.ToString(),
             std::string("max 1.000000 * x1 + 1.000000 * x2 + -0.000000\n"
                         "-1.000000 * base0 + -2.000000 * x1 + -1.000000 * x2 + 12.000000 = 0.000000\n"
                         "-1.000000 * base1 + -1.000000 * x1 + -2.000000 * x2 + 9.000000 = 0.000000\n"));
  }
  {
    LPModel model = Load(dir, "test4.txt");
    model.ToStandardForm();
    model.ToSlackForm();
    CHECK_EQ(model.ToString(),
             std::string("max -6.000000 * x1 + -3.000000 * x2 + -0.000000\n"
                         "-1.000000 * base0 + 1.000000 * x1 + 1.000000 * x2 + -1.000000 = 0.000000\n"
                         "-1.000000 * base1 + 2.000000 * x1 + -1.000000 * x2 + -1.000000 = 0.000000\n"
                         "-1.000000 * base2 + -3.000000 * x2 + 2.000000 = 0.000000\n"));
  }
}

static bool Near(Num a, double b) { return std::fabs(a.float_value - b) <= 1e-6f; }

static void TableauCases(const std::string& dir) {
  {  // solve of test3: optimum -7 at x1 = 5, x2 = 2
    LPModel model = Load(dir, "test3.txt");
    model.ToStandardForm();
    model.ToSlackForm();
    model.ToTableau();
    CHECK_EQ(model.TableauSimplexSolve(), SOLVED);
    CHECK_EQ(model.GetTableauSimplexOptimum() == Num(-7.0f), true);
    auto sol = model.GetTableauSimplexSolution();
    CHECK_EQ(sol.size(), (size_t)2);
    CHECK_EQ(sol[Variable("x1")] == Num(5.0f) && sol[Variable("x2")] == Num(2.0f), true);
  }
  {  // phase 1 on test4, then the solve
    LPModel model = Load(dir, "test4.txt");
    model.ToStandardForm();
    model.ToSlackForm();
    model.ToTableau();
    CHECK_EQ(model.TableauSimplexInitialize(), SOLVED);
    CHECK_EQ(model.PrintTableau(),
             std::string("-4.000000 * base0 + -1.000000 * base1 + -5.000000\n"
                         "0.333333 * base0 + 0.333333 * base1 + -1.000000 * x1 + 0.666667\n"
                         "0.666667 * base0 + -0.333333 * base1 + -1.000000 * x2 + 0.333333\n"
                         "-2.000000 * base0 + 1.000000 * base1 + -1.000000 * base2 + 1.000000\n"));
    CHECK_EQ(model.TableauSimplexSolve(), SOLVED);
    CHECK_EQ(model.GetTableauSimplexOptimum() == Num(5.0f), true);
    auto sol = model.GetTableauSimplexSolution();
    CHECK_EQ(sol.size(), (size_t)2);
    CHECK_EQ(Near(sol[Variable("x1")], 2.0f / 3) && Near(sol[Variable("x2")], 1.0f / 3), true);
  }
  // reference lp_revised_simplex_tableau_test.cc:7-98
  struct RevisedCase { const char* file; bool check_opt; float opt; double x1, x2; };
  const RevisedCase revised[] = {{"test3.txt", true, -7.0f, 5.0, 2.0}, {"test8.txt", false, 0.0f, 4.0, 2.0},
                                 {"test5.txt", true, 23.75f, 3.75, 1.25}, {"test4.txt", true, 5.0f, 2.0f / 3, 1.0f / 3}};
  for (const auto& c : revised) {
    LPModel model = Load(dir, c.file);
    model.ToStandardForm();
    model.ToSlackForm();
    model.ToTableau(COLUMN_ONLY);
    CHECK_EQ(model.TableauRevisedSimplexSolve(), SOLVED);
    if (c.check_opt) CHECK_EQ(model.GetTableauRevisedSimplexOptimum() == Num(c.opt), true);
    auto sol = model.GetTableauRevisedSimplexSolution();
    CHECK_EQ(sol.size(), (size_t)2);
    CHECK_EQ(Near(sol[Variable("x1")], c.x1) && Near(sol[Variable("x2")], c.x2), true);
  }
  {  // ToTableau snapshot of test9
    LPModel model = Load(dir, "test9.txt");
    model.ToStandardForm();
    model.ToSlackForm();
    model.ToTableau();
    CHECK_EQ(model.PrintTableau(),
             std::string("-0.800000 * x1 + -0.500000 * x2 + -0.900000 * x3 + -1.500000 * x4\n"
                         "-1.000000 * base0 + 1000.000000 * x1 + 1500.000000 * x2 + 1750.000000 * x3 + 3250.000000 * x4 + -4000.000000\n"
                         "-1.000000 * base1 + 0.600000 * x1 + 0.270000 * x2 + 0.680000 * x3 + 0.300000 * x4 + -1.000000\n"
                         "-1.000000 * base2 + 17.500000 * x1 + 7.700000 * x2 + 30.000000 * x4 + -30.000000\n"));
  }
  {  // two explicit pivots on test3
    LPModel model = Load(dir, "test3.txt");
    model.ToStandardForm();
    model.ToSlackForm();
    model.ToTableau();
    model.TableauPivot(Variable("base0"), Variable("x1"), 0);
    CHECK_EQ(model.PrintTableau(),
             std::string("-0.500000 * base0 + 0.500000 * x2 + 6.000000\n"
                         "-0.500000 * base0 + -1.000000 * x1 + -0.500000 * x2 + 6.000000\n"
                         "0.500000 * base0 + -1.000000 * base1 + -1.500000 * x2 + 3.000000\n"));
    model.TableauPivot(Variable("base1"), Variable("x2"), 1);
    CHECK_EQ(model.PrintTableau(),
             std::string("-0.333333 * base0 + -0.333333 * base1 + 7.000000\n"
                         "-0.666667 * base0 + 0.333333 * base1 + -1.000000 * x1 + 5.000000\n"
                         "0.333333 * base0 + -0.666667 * base1 + -1.000000 * x2 + 2.000000\n"));
  }
}

int main(int argc, char** argv) {
  bool gpu = false;
  std::string dir;
  for (int i = 1; i < argc; i++) {
    if (std::strcmp(argv[i], "--gpu") == 0) gpu = true;
    else dir = argv[i];
  }
  if (dir.empty()) { std::cerr << "usage: " << argv[0] << " [--gpu] <fixture-dir>\n"; return 2; }
  HostCases(dir);
  if (gpu) TableauCases(dir);
  std::cout << g_checked << " checks, " << g_failed << " failed" << (gpu ? " (host + tableau-on-GPU cases)" : " (host cases)") << "\n";
  return g_failed ? 1 : 0;
}
