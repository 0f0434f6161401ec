// Netlib ingestion, first stage: the compressed form in which netlib ships its LPs
// (reference datasets/netlib/<name>, expanded upstream by tools/emps.c, tools/preprocessing.sh:2-9)
// -> fixed-format MPS text, byte for byte what `emps <file>` prints with its default options
// (two entries per line, mystery lines passed through).  mps_text.h takes it from there.
#pragma once
#include <istream>
#include <string>

// Throws std::runtime_error on a truncated file, a checksum mismatch or a malformed field;
// `label` names the input in the message.
std::string ExpandNetlib(std::istream& in, const std::string& label);

// True when the stream holds the compressed form: its first NAME line is followed by the line of
// eight problem statistics (plain MPS continues with ROWS).  Leaves the stream rewound.
bool LooksLikeCompressedNetlib(std::istream& in);
