// Behavioural contract: reference tools/emps.c (David M. Gay's expander for netlib's compressed
// LPs) with default options; function by function:
//   line layer + checksums   emps.c:204-283   (one check symbol per line, a checksum line every 71)
//   supersparse index        emps.c:299-321
//   number field             emps.c:327-436
//   problem layout           emps.c:472-573
//   entry sections           emps.c:579-647
#include "netlib_text.h"

#include <cstdio>
#include <stdexcept>
#include <vector>

namespace {

// The 92 printable symbols of the encoding, in digit order (emps.c:47-48).
const char kSymbols[] =
    "!\"#$%&'()*+,-./0123456789;<=>?@"
    "ABCDEFGHIJKLMNOPQRSTUVWXYZ[]^_`abcdefghijklmnopqrstuvwxyz{|}~";
constexpr int kRadix = 92;           // digits of a general number; also the value of a foreign character
constexpr int kHalf = 46;            // digits of an index; a digit >= 46 closes a multi-digit index
constexpr int kQuarter = 23;
constexpr size_t kLinesPerCheck = 71;
constexpr long kFieldLimit = 1L << 52;   // no index or integer of a real file comes near; 46 times this still fits a long
const char* const kBoundKinds[] = {"UP", "LO", "FX", "FR", "MI", "PL"};

std::string Clip(const std::string& s, size_t width) { return s.substr(0, std::min(width, s.find('\0') == std::string::npos ? s.size() : s.find('\0'))); }
std::string PadRight(std::string s, size_t width) {
  if (s.size() < width) s.append(width - s.size(), ' ');
  return s;
}
std::string Name8(const std::string& s) { return Clip(s, 8); }

class Expander {
 public:
  Expander(std::istream& in, const std::string& label) : in_(in), label_(label) {
    for (int& v : value_of_) v = kRadix;
    for (int d = 0; kSymbols[d]; d++) value_of_[(unsigned char)kSymbols[d]] = d;
  }

  std::string Run() {
    NextLine();
    for (;;) {
      check_.clear();
      while (line_.compare(0, 4, "NAME") != 0)
        if (!NextLine()) return out_;
      may_end_ = false;
      out_ += line_ + "\n";
      long ncol, colmax, nz, nrhs, rhsnz, nran, ranz, nbd, bdnz, ns;
      NextLine();
      const bool ok1 = std::sscanf(line_.c_str(), "%ld %ld %ld %ld %ld %ld %ld %ld", &nrow_, &ncol, &colmax, &nz, &nrhs, &rhsnz, &nran, &ranz) == 8;
      if (ok1) NextLine();
      if (!ok1 || std::sscanf(line_.c_str(), "%ld %ld %ld", &nbd, &bdnz, &ns) != 3) Fail("bad statistics line: " + line_);
      if (nrow_ < 0 || ncol < 0 || ns < 0 || ns > (1L << 26) || nrow_ + ncol > (1L << 26)) Fail("problem too big");
      check_.clear();   // the NAME and statistics lines are outside the checksums
      names_.assign(size_t(nrow_ + ncol) + 1, std::string());
      next_name_ = nrow_;

      numbers_.assign(1, std::string());   // 1-based
      table_open_ = true;
      Discard();
      for (long i = 0; i < ns; i++) {
        Refill();
        numbers_.push_back(Number());
      }
      table_open_ = false;

      for (long i = 1; i <= nrow_; i++) {
        NextLine();
        if (i == 1) out_ += "ROWS\n";
        if (line_.empty()) Fail("empty row line");
        out_ += std::string(" ") + line_[0] + "  " + line_.substr(1) + "\n";
        names_[i] = Name8(line_.substr(1));
      }
      Section("COLUMNS", nz, 1);
      Section("RHS", rhsnz, 2);
      Section("RANGES", ranz, 3);
      Section("BOUNDS", bdnz, 4);
      if (!check_.empty()) VerifyCheckLine();
      out_ += "ENDATA\n";
      may_end_ = true;   // another problem may follow in the same file
      check_.clear();
      if (!NextLine()) return out_;
    }
  }

 private:
  [[noreturn]] void Fail(const std::string& what) const {
    throw std::runtime_error("netlib expansion of " + label_ + ": " + what + " (line " + std::to_string(line_no_) + ")");
  }

  // ---- lines and checksums -------------------------------------------------------------------------------
  char CheckSymbol(const std::string& s) const {
    unsigned x = 0;
    for (unsigned char c : s) x = (x >> 1) + value_of_[c] + ((x & 1) ? 16384u : 0u);
    return kSymbols[x % kRadix];
  }

  bool ReadRaw(std::string& s) {
    line_no_++;
    if (!std::getline(in_, s)) return false;
    if (s.size() > 75) Fail("line longer than 75 characters");
    return true;
  }

  // The next data line (mystery lines, which start with ':', are passed through); false at a permitted end.
  bool NextLine() {
    for (;;) {
      if (!ReadRaw(line_)) {
        if (may_end_) return false;
        Fail("premature end of file");
      }
      check_.push_back(CheckSymbol(line_));
      if (check_.size() >= kLinesPerCheck) VerifyCheckLine();
      if (!line_.empty() && line_[0] == ':') {
        out_ += line_.substr(1) + "\n";
        continue;
      }
      pos_ = 0;
      return true;
    }
  }

  // The file carries, after every 71 lines and at the end of a problem, a blank followed by the check
  // symbols of the lines since the last such line.
  void VerifyCheckLine() {
    std::string got;
    for (;;) {
      if (!ReadRaw(got)) Fail("premature end of file");
      if (got == " " + check_) break;
      if (!got.empty() && got[0] == ':' && check_.size() < kLinesPerCheck) {   // a mystery line in front of it
        check_.push_back(CheckSymbol(got));
        out_ += got.substr(1) + "\n";
        continue;
      }
      Fail("check sum error: expected `" + check_ + "`, got `" + got + "`");
    }
    check_.clear();
  }

  // ---- fields ------------------------------------------------------------------------------------------------
  bool LineSpent() const { return pos_ >= line_.size(); }
  void Discard() { pos_ = line_.size(); }
  void Refill() {
    if (LineSpent()) NextLine();
  }
  int Take() {
    if (LineSpent()) Fail("field runs past the end of its line");
    const int v = value_of_[(unsigned char)line_[pos_++]];
    if (v == kRadix) Fail("foreign character in a field");
    return v;
  }

  // One digit 23..45 is the index 0..22; otherwise digits to base 46, the last one offset by 46.
  long Index() {
    int k = Take();
    if (k >= kHalf) Fail("bad index");
    if (k >= kQuarter) return k - kQuarter;
    long x = k;
    for (;;) {
      k = Take();
      if (x > kFieldLimit) Fail("index field too long");   // a crafted line must not overflow the accumulator
      x = x * kHalf + k;
      if (k >= kHalf) return x - kHalf;
    }
  }

  // A number as MPS text, right-justified to 12 characters: an index into the table of repeated numbers,
  // an integer, or mantissa digits with a decimal exponent.
  std::string Number() {
    const size_t start = pos_;
    int k = Take();
    if (k < kHalf) {
      pos_ = start;
      const long idx = Index();
      if (table_open_ || idx < 1 || idx >= (long)numbers_.size()) Fail("index " + std::to_string(idx) + " outside the number table");
      return numbers_[idx];
    }
    k -= kHalf;
    const bool negative = k >= kQuarter;
    if (negative) k -= kQuarter;
    const int plain_limit = negative ? 11 : 12;   // longest plain notation before the exponent form takes over
    std::string body;
    if (k >= 11) {   // integer
      k -= 11;
      long x;
      if (k >= 6) x = k - 6;
      else {
        x = k;
        for (;;) {
          k = Take();
          if (x > kFieldLimit) Fail("integer field too long");
          x = x * kHalf + k;
          if (k >= kHalf) { x -= kHalf; break; }
        }
      }
      body = std::to_string(x) + ".";
    } else {         // k + 1 further mantissa digits to base 92, split in two words when the first fills up
      int ex = Take() - 50;
      long x = Take(), y = 0;
      while (--k >= 0) {
        if (x >= 100000000) { y = x; x = Take(); }
        else x = x * kRadix + Take();
      }
      std::string digits;
      if (y) {
        digits = std::to_string(y);
        std::string low;   // the second word carries a leading 1 that is not a digit
        for (long t = x; t > 1; t /= 10) low.insert(low.begin(), char('0' + t % 10));
        digits += low;
      } else {
        digits = std::to_string(x);
      }
      const int nd = (int)digits.size() + ex;   // digits in front of the decimal point
      bool exponent_form = false;
      if (ex > 0) {
        if (nd < plain_limit || ex < 3) body = digits + std::string(ex, '0') + ".";
        else exponent_form = true;
      } else if (nd >= 0) {
        body = digits.substr(0, nd) + "." + digits.substr(nd);
      } else if (ex > -plain_limit) {
        body = "." + std::string(-nd, '0') + digits;
      } else {
        exponent_form = true;
      }
      if (exponent_form) {
        int e10 = ex + (int)digits.size() - 1;
        size_t lead = 0;
        if (e10 == -10) e10 = -9;   // saves the second exponent digit: .ddddE-9
        else {
          if (e10 > 9 && e10 <= (int)digits.size() + 8) { lead = e10 - 9; e10 = 9; }   // likewise ddd.ddE9
          lead++;
        }
        body = digits.substr(0, lead) + "." + digits.substr(lead) + "E";
        if (e10 < 0) { body += "-"; e10 = -e10; }
        if (e10) body += std::to_string(e10);
      }
    }
    if (negative) body.insert(body.begin(), '-');
    if (body.size() < 12) body.insert(0, 12 - body.size(), ' ');
    return body;
  }

  const std::string& FetchName(long i) const {
    if (i <= 0 || i >= (long)names_.size()) Fail("name index " + std::to_string(i) + " out of range");
    return names_[i];
  }

  // COLUMNS (kind 1), RHS (2), RANGES (3), BOUNDS (4): `count` entries; index 0 announces the name of the
  // next column (or right-hand-side / range / bound set) on the rest of its line.
  void Section(const char* head, long count, int kind) {
    if (!count) {
      if (kind <= 2) out_ += std::string(head) + "\n";
      return;
    }
    bool first = true;
    int held = 0;   // entries waiting for their partner on the same output line
    std::string column, name[2], value[2];
    auto flush_single = [&] { out_ += "    " + PadRight(column, 8) + "  " + PadRight(name[0], 8) + "  " + value[0].substr(0, 15) + "\n"; };
    Discard();
    while (count--) {
      Refill();
      if (first) {
        out_ += std::string(head) + "\n";
        first = false;
      }
      long n;
      while (!(n = Index())) {
        if (held) { flush_single(); held = 0; }
        column = Name8(line_.substr(pos_));
        if (kind == 1) {
          if (++next_name_ >= (long)names_.size()) Fail("more columns than announced");
          names_[next_name_] = column;
        }
        NextLine();
      }
      if (kind >= 4) {
        if (n < 1 || n >= 7) Fail("bad bound type index = " + std::to_string(n));
        Refill();
        name[0] = FetchName(nrow_ + Index());
        if (n-- >= 4) {   // FR, MI, PL carry no value
          out_ += std::string(" ") + kBoundKinds[n] + " " + PadRight(column, 8) + "  " + name[0] + "\n";
          continue;
        }
      } else {
        name[held] = FetchName(n);
      }
      Refill();
      value[held] = Number();
      if (kind <= 3) {
        if (++held == 1) continue;
        out_ += "    " + PadRight(column, 8) + "  " + PadRight(name[0], 8) + "  " + PadRight(value[0].substr(0, 15), 15) +
                PadRight(name[1], 8) + "  " + value[1].substr(0, 15) + "\n";
        held = 0;
      } else {
        out_ += std::string(" ") + kBoundKinds[n] + " " + PadRight(column, 8) + "  " + PadRight(name[0], 8) + "  " + value[0].substr(0, 15) + "\n";
      }
    }
    if (held) flush_single();
  }

  std::istream& in_;
  std::string label_;
  int value_of_[256];
  long line_no_ = 0;
  bool may_end_ = false;
  std::string check_;   // check symbols of the lines since the last checksum line
  std::string line_;
  size_t pos_ = 0;
  std::string out_;
  long nrow_ = 0, next_name_ = 0;
  bool table_open_ = false;
  std::vector<std::string> numbers_;   // table of repeated numbers, 1-based
  std::vector<std::string> names_;     // rows 1..nrow, then columns in order of appearance
};

}  // namespace

std::string ExpandNetlib(std::istream& in, const std::string& label) { return Expander(in, label).Run(); }

bool LooksLikeCompressedNetlib(std::istream& in) {
  std::string line;
  long v[8];
  bool yes = false;
  while (std::getline(in, line)) {   // the expander skips whatever precedes the NAME line, so does this test
    if (line.compare(0, 4, "NAME") != 0) continue;
    yes = std::getline(in, line) &&
          std::sscanf(line.c_str(), "%ld %ld %ld %ld %ld %ld %ld %ld", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7) == 8;
    break;
  }
  in.clear();
  in.seekg(0);
  return yes;
}
