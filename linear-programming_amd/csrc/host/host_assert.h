// Assertions that stay live in optimized builds: the reference strips -DNDEBUG
// from its release flags on purpose (reference CMakeLists.txt:12), so violated
// invariants abort there too.
#pragma once
#include <cstdio>
#include <cstdlib>

#define LP_ASSERT(cond)                                                                  \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      std::fprintf(stderr, "%s:%d: assertion `%s' failed\n", __FILE__, __LINE__, #cond); \
      std::abort();                                                                      \
    }                                                                                    \
  } while (0)
