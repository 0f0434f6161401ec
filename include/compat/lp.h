/* Source-compatibility shim for the reference's lp.h (LPModel, Model, Result,
 * PivotingStrategy, TableauStorageFormat, tableau_index_t) -- tableau path only. */
#pragma once
#include "lp_model.h"
