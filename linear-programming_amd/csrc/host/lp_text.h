// Text front-end: lexer and line parser for the LP description format
//   max|min <expression> \n st \n <constraint> ... [\n a, b, c  (integer variables)]
// Same types and entry points as the reference's parser.h (Token, Lexer::Scan,
// Parser::Parse x3, ParseOptimizationObject/Constraint/Expression).
#pragma once

#include <fstream>

#include "algebra.h"
#include "lp_model.h"

struct Token {
  enum TokenType {
    UNKNOWN = 0,
    VAR,  // [a-zA-Z_][a-zA-Z_0-9]*
    NUM,
    MAX, MIN, ST,        // reserved words
    EQ, GE, LE,          // relations
    ADD, SUB, MUL,       // arithmetic (no division)
    EOL,                 // end of line
    COMMA,
  };
  TokenType type;
  std::string lexim;
  int start_index;
  int end_index;
};

inline bool operator==(const Token a, const Token b) {
  return a.type == b.type && a.lexim == b.lexim && a.start_index == b.start_index && a.end_index == b.end_index;
}
inline bool operator!=(const Token a, const Token b) { return !(a == b); }

class Lexer {
 public:
  std::vector<Token> Scan(std::string input);

 private:
  bool IsLetter(char c);
  bool IsDigit(char c);
};

class Parser {
 public:
  Model Parse(std::vector<Token> tokens);
  Model Parse(std::string input);
  Model Parse(std::ifstream& file);

  OptimizationObject ParseOptimizationObject(std::vector<Token> tokens);
  Constraint ParseConstraint(std::vector<Token> tokens);
  Expression ParseExpression(std::vector<Token> tokens);
};
