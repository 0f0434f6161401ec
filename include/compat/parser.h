/* Source-compatibility shim for the reference's parser.h (Token, Lexer, Parser). */
#pragma once
#include "lp_text.h"
