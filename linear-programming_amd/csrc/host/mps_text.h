// Netlib ingestion: fixed-format MPS file -> the LP text format of lp_text.h,
// written the way the reference's tools/mps_load.py writes it (variables renamed
// x0, x1, ... in order of first appearance, "%f" numbers, bound lines last), so
// that the parser sees byte-identical input and the tableau matches the
// reference's datasets/netlib/<name>.txt.
#pragma once
#include <istream>
#include <string>

// `mps_path` may also hold netlib's compressed form (netlib_text.h), which is expanded first.
std::string MpsToLpText(const std::string& mps_path);
std::string MpsStreamToLpText(std::istream& mps);
