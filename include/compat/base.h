/* Source-compatibility shim: code written against the reference's base.h
 * (Variable, Num, Expression, Constraint, OptimizationObject, Timer) builds
 * against this tree unchanged. */
#pragma once
#include "algebra.h"
