// lp_host_probe <mode> <file>: prints intermediate states of the host layer so
// that CPU-only tests can compare them with the strings the reference's own
// unit tests expect (parser_test.cc, lp_test.cc) and with the oracle's tableau.
//   tokens | parse | standard | slack | tableau | mps2txt | expand
#include <cinttypes>
#include <cstring>
#include <fstream>
#include <iostream>

#include "lp_model.h"
#include "lp_text.h"
#include "mps_text.h"
#include "netlib_text.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cerr << "usage: " << argv[0] << " tokens|parse|standard|slack|tableau|mps2txt|expand <file>\n";
    return 2;
  }
  const std::string mode = argv[1], path = argv[2];
  if (mode == "expand") {   // netlib's compressed form -> MPS text
    std::ifstream f(path);
    if (!f.is_open()) { std::cerr << "cannot open " << path << "\n"; return 2; }
    try {
      std::cout << ExpandNetlib(f, path);
    } catch (const std::exception& ex) {
      std::cerr << ex.what() << "\n";
      return 3;
    }
    return 0;
  }
  if (mode == "mps2txt") {
    std::cout << MpsToLpText(path);
    return 0;
  }
  std::string text;
  if (path.size() > 4 && path.substr(path.size() - 4) == ".mps") {
    text = MpsToLpText(path);
  } else {
    std::ifstream f(path);
    if (!f.is_open()) { std::cerr << "cannot open " << path << "\n"; return 2; }
    std::string line;
    while (std::getline(f, line)) text += line + "\n";
  }
  if (mode == "tokens") {
    Lexer lexer;
    for (const Token& tk : lexer.Scan(text))
      std::cout << (int)tk.type << " " << tk.start_index << " " << tk.end_index << " " << (tk.type == Token::EOL ? "\\n" : tk.lexim) << "\n";
    return 0;
  }
  Parser parser;
  LPModel model(parser.Parse(text));
  if (mode == "parse") { std::cout << model.ToString(); return 0; }
  model.ToStandardForm();
  if (mode == "standard") {
    std::cout << model.ToString();
    std::cout << "sanity " << StandardFormSanityCheck(model) << "\n";
    for (auto& e : model.GetRawVariableExpression()) std::cout << "raw " << e.first.ToString() << " = " << e.second.ToString() << "\n";
    return 0;
  }
  model.ToSlackForm();
  if (mode == "slack") {
    std::cout << model.ToString();
    std::cout << "sanity " << SlackFormSanityCheck(model) << "\n";
    return 0;
  }
  if (mode == "tableau") {   // bit patterns, column-major: "m N" then one line per column, then obj, then flags
    HostTableau h = model.BuildHostTableau();
    std::cout << h.m << " " << h.N << "\n";
    auto hex = [](double v) { uint64_t u; std::memcpy(&u, &v, 8); char b[20]; std::snprintf(b, sizeof(b), "%016" PRIx64, u); return std::string(b); };
    for (int64_t j = 0; j < h.N + 2; j++) {
      for (int64_t i = 0; i < h.m; i++) std::cout << (i ? " " : "") << hex(h.T[(size_t)j * h.m + i]);
      std::cout << "\n";
    }
    for (int64_t j = 0; j < h.N + 2; j++) std::cout << (j ? " " : "") << hex(h.obj[j]);
    std::cout << "\n";
    for (int64_t j = 0; j < h.N + 2; j++) std::cout << (j ? " " : "") << (int)h.is_basic[j];
    std::cout << "\n";
    return 0;
  }
  std::cerr << "unknown mode " << mode << "\n";
  return 2;
}
