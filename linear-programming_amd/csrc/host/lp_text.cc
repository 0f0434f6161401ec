// Lexer / parser of the LP text format.  Behavioural contract: reference
// parser.cc (token boundaries and indices :3-122, number literals through
// std::stof :179,188, constraint = lhs - rhs :141-167, integer line :238-275).
#include "lp_text.h"

#include "host_assert.h"

bool Lexer::IsLetter(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
bool Lexer::IsDigit(char c) { return c >= '0' && c <= '9'; }

std::vector<Token> Lexer::Scan(std::string input) {
  std::vector<Token> out;
  const int n = (int)input.size();
  bool open = false;           // a VAR/NUM token is being accumulated
  Token cur{Token::UNKNOWN, "", -1, -1};
  auto flush = [&](int end) {
    if (!open) return;
    cur.lexim = input.substr(cur.start_index, end - cur.start_index);
    cur.end_index = end;
    out.push_back(cur);
    open = false;
  };
  auto start = [&](Token::TokenType t, int at) {
    cur = Token{t, "", at, -1};
    open = true;
  };
  auto symbol = [&](Token::TokenType t, const char* text, int at, int len) { out.push_back(Token{t, text, at, at + len}); };

  for (int i = 0; i < n; i++) {
    const char c = input[i];
    if (c == '\n') {
      flush(i);
      symbol(Token::EOL, "\n", i, 1);
    } else if (c == '\0' || c == ' ' || c == '\t') {
      flush(i);
    } else if (IsLetter(c)) {
      if (!open) start(Token::VAR, i);
      else if (cur.type == Token::NUM) throw std::runtime_error("cannot concat num + var");
    } else if (IsDigit(c)) {
      if (!open) start(Token::NUM, i);
    } else if (c == '.') {
      if (!open) start(Token::NUM, i);
      else LP_ASSERT(cur.type == Token::NUM);
    } else if (c == '+' || c == '-' || c == '*' || c == '>' || c == '<' || c == '=' || c == ',') {
      flush(i);
      switch (c) {
        case '+': symbol(Token::ADD, "+", i, 1); break;
        case '-': symbol(Token::SUB, "-", i, 1); break;
        case '*': symbol(Token::MUL, "*", i, 1); break;
        case '=': symbol(Token::EQ, "=", i, 1); break;
        case ',': symbol(Token::COMMA, ",", i, 1); break;
        default:  // '>' or '<' must be followed by '='
          LP_ASSERT(i + 1 < n);
          LP_ASSERT(input[i + 1] == '=');
          symbol(c == '>' ? Token::GE : Token::LE, c == '>' ? ">=" : "<=", i, 2);
          i += 1;
      }
    } else {
      throw std::runtime_error("Lexing error: un-recognized symbol");
    }
  }
  flush(n);
  for (Token& tk : out) {
    if (tk.type != Token::VAR) continue;
    if (tk.lexim == "MAX" || tk.lexim == "max") tk.type = Token::MAX;
    if (tk.lexim == "MIN" || tk.lexim == "min") tk.type = Token::MIN;
    if (tk.lexim == "ST" || tk.lexim == "st") tk.type = Token::ST;
  }
  return out;
}

OptimizationObject Parser::ParseOptimizationObject(std::vector<Token> tokens) {
  LP_ASSERT(!tokens.empty());
  LP_ASSERT(tokens[0].type == Token::MAX || tokens[0].type == Token::MIN);
  OptimizationObject obj(FLOAT);
  obj.opt_type = tokens[0].type == Token::MAX ? OptimizationObject::MAX : OptimizationObject::MIN;
  obj.expression = ParseExpression(std::vector<Token>(tokens.begin() + 1, tokens.end()));
  return obj;
}

Constraint Parser::ParseConstraint(std::vector<Token> tokens) {
  LP_ASSERT(tokens.size() >= 3);
  int rel = -1;
  for (int i = 0; i < (int)tokens.size(); i++) {
    const auto t = tokens[i].type;
    if (t != Token::EQ && t != Token::GE && t != Token::LE) continue;
    if (rel >= 0) throw std::runtime_error("More than one compare symbol found in one constraint");
    rel = i;
  }
  LP_ASSERT(rel >= 0);
  Constraint con(FLOAT);
  con.equation_type = tokens[rel].type == Token::EQ ? Constraint::EQ : (tokens[rel].type == Token::GE ? Constraint::GE : Constraint::LE);
  con.expression = ParseExpression(std::vector<Token>(tokens.begin(), tokens.begin() + rel));
  con.expression -= ParseExpression(std::vector<Token>(tokens.begin() + rel + 1, tokens.end()));
  return con;
}

namespace {
// One signed term: NUM | VAR | NUM * VAR.  Literals go through std::stof, so
// every coefficient is a float widened to double.
Expression ParseTerm(const std::vector<Token>& term) {
  LP_ASSERT(term.size() == 1 || term.size() == 3);
  if (term.size() == 1) {
    LP_ASSERT(term[0].type == Token::NUM || term[0].type == Token::VAR);
    if (term[0].type == Token::NUM) return Expression(Num(std::stof(term[0].lexim)));
    return Expression(Variable(term[0].lexim));
  }
  LP_ASSERT(term[0].type == Token::NUM);
  LP_ASSERT(term[1].type == Token::MUL);
  LP_ASSERT(term[2].type == Token::VAR);
  return Num(std::stof(term[0].lexim)) * Variable(term[2].lexim);
}
}  // namespace

Expression Parser::ParseExpression(std::vector<Token> tokens) {
  Expression acc = Expression(kFloatZero);
  std::vector<Token> term;
  bool positive = true;
  auto commit = [&]() {
    Expression e = ParseTerm(term);
    if (positive) acc += e; else acc -= e;
  };
  for (size_t i = 0; i < tokens.size(); i++) {
    const Token& tk = tokens[i];
    if (tk.type == Token::ADD || tk.type == Token::SUB) {
      if (i != 0) commit();
      positive = tk.type == Token::ADD;
      term.clear();
    } else {
      term.push_back(tk);
    }
  }
  if (!term.empty()) commit();
  return acc;
}

Model Parser::Parse(std::vector<Token> tokens) {
  std::vector<std::vector<Token>> lines(1);
  for (const Token& tk : tokens) {
    if (tk.type == Token::EOL) lines.emplace_back();
    else lines.back().push_back(tk);
  }
  if (lines.back().empty()) lines.pop_back();
  LP_ASSERT(lines.size() >= 2);
  LP_ASSERT(lines[0].size() >= 2);
  LP_ASSERT(lines[1].size() == 1);
  LP_ASSERT(lines[1][0].type == Token::ST);
  OptimizationObject opt_obj = ParseOptimizationObject(lines[0]);

  // A last line containing commas lists the integer variables.
  size_t n_lines = lines.size();
  std::set<Variable> integers;
  bool has_integer_line = false;
  for (const Token& tk : lines.back()) has_integer_line |= tk.type == Token::COMMA;
  if (has_integer_line) {
    n_lines -= 1;
    for (const Token& tk : lines.back())
      if (tk.type == Token::VAR) integers.insert(Variable(tk.lexim, INTEGER));
  }
  auto retag = [&](Expression& expr) {
    for (const Variable& iv : integers) {
      Variable fv(iv.variable_name, FLOAT);
      if (expr.GetCoeffOf(fv) != kFloatZero) {
        expr.SetCoeffOf(iv, expr.GetCoeffOf(fv));
        expr.SetCoeffOf(fv, kFloatZero);
      }
    }
  };
  retag(opt_obj.expression);
  std::vector<Constraint> constraints;
  for (size_t i = 2; i < n_lines; i++) {
    Constraint con = ParseConstraint(lines[i]);
    retag(con.expression);
    constraints.push_back(con);
  }
  return {constraints, opt_obj};
}

Model Parser::Parse(std::string input) {
  Lexer lexer;
  return Parse(lexer.Scan(input));
}

Model Parser::Parse(std::ifstream& file) {
  if (!file.is_open()) throw std::runtime_error("Cannot find file\n");
  std::string text, line;
  while (std::getline(file, line)) text += line + "\n";
  file.close();
  return Parse(text);
}
