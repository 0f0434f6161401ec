// Netlib ingestion: fixed-format MPS file -> the LP text format of lp_text.h,
// written the way the reference's tools/mps_load.py writes it (variables renamed
// x0, x1, ... in order of first appearance, "%f" numbers, bound lines last), so
// that the parser sees byte-identical input and the tableau matches the
// reference's datasets/netlib/<name>.txt.
#pragma once
#include <string>

std::string MpsToLpText(const std::string& mps_path);
