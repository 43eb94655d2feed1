// sgf_record.h -- host-side SGF ingest for the offline (SGF replay) path.
//
// Produces, for one SGF text, the flat list of moves the reference's Sgf::iterator walks
// (go/sgf.h:130-177) as board actions (x*19+y, -1 pass), plus the header fields the loader
// uses.  The observable behaviour of the reference parser (go/sgf.cc) is kept, including its
// quirks, because batch parity depends on them:
//   * the first node is the header; its keys are trimmed before comparison (sgf.cc:125-158);
//   * every later ';' starts a node, wherever it stands: variations are flattened in file order
//     (sgf.cc:196-225 skips to the next ';' and chains nodes as siblings);
//   * inside a node a key is the RAW text between the previous ']' (or the node start) and '[';
//     only a raw key of exactly one character B / W / C is interpreted (sgf.cc:177-194), so
//     "\nB[dd]" in the middle of a node is not a move;
//   * a backslash skips the next character (sgf.cc:87-88);
//   * a move value shorter than two characters or off the 19x19 board is a pass
//     (str2coord, go/sgf.h:23-32); values are left-trimmed of ' ' and '\n' only;
//   * NumMoves counts nodes up to the first pass (sgf.cc:45-51).
// A node that carries no B/W property has an indeterminate move in the reference (SgfEntry is
// never initialised, go/sgf.h:63-74); here it is a pass.
#pragma once

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace gogame {

struct SgfRecord {
  bool ok = false;            // header found and at least one move node
  int board_size = 19;        // SZ
  int handi = 0;              // HA
  float komi = 7.5f;          // KM
  int winner = 3;             // 1 black, 2 white, 3 unknown (S_OFF_BOARD, go/sgf.h:100)
  std::vector<int16_t> moves; // every node after the header, as actions (-1 pass)
  std::vector<uint8_t> players;  // 1 black, 2 white, 0 none, per node
  int num_moves = 0;          // Sgf::NumMoves()
  // How many of `moves` GoState::forward accepts when the record is replayed from its start (handicap
  // placed): a property of the record, found by one device replay when a loader first uses it
  // (DeviceEnv::count_legal_prefix); -1 = not yet known.  Host bookkeeping only.
  mutable int legal_prefix = -1;
};

namespace detail {

// trim of go/sgf.cc:15-22: leading ' ' / '\n' removed; the length argument of its substr is
// r + 1 (not r - l + 1), so trailing blanks survive whenever anything was cut in front.
inline std::string ref_trim(const std::string& s) {
  int l = 0;
  while (l < (int)s.size() && (s[l] == ' ' || s[l] == '\n')) ++l;
  int r = (int)s.size() - 1;
  while (r >= 0 && (s[r] == ' ' || s[r] == '\n')) --r;
  if (l > (int)s.size()) return std::string();
  return s.substr(l, r + 1 < 0 ? 0 : r + 1);
}

inline int action_of(const std::string& v) {
  if (v.size() < 2) return -1;
  const int x = v[0] - 'a', y = v[1] - 'a';
  if (x < 0 || x >= 19 || y < 0 || y >= 19) return -1;
  return x * 19 + y;
}

// Scans the key[value] pairs of one node starting at `i`; calls cb(raw_key, raw_value) for each
// and returns the offset of the ';' or ')' that ends the node (or `end`).
template <typename F>
inline int scan_node(const std::string& t, int i, int end, F cb) {
  int start = i;
  bool in_value = false, esc = false;
  int key_b = 0, key_e = 0;
  for (; i < end; ++i) {
    const char c = t[i];
    if (c == '\\') { esc = !esc; continue; }
    if (esc) { esc = false; continue; }
    if (!in_value) {
      if (c == '[') { key_b = start; key_e = i; start = i + 1; in_value = true; }
      else if (c == ';' || c == ')') return i;
    } else if (c == ']') {
      cb(t.substr(key_b, key_e - key_b), t.substr(start, i - start));
      start = i + 1;
      in_value = false;
    }
  }
  return i;
}

inline bool to_int(const std::string& v, int* out) {
  char* e = nullptr;
  const long x = std::strtol(v.c_str(), &e, 10);
  if (e == v.c_str()) return false;
  *out = (int)x;
  return true;
}

inline bool to_float(const std::string& v, float* out) {
  char* e = nullptr;
  const float x = std::strtof(v.c_str(), &e);
  if (e == v.c_str()) return false;
  *out = x;
  return true;
}

}  // namespace detail

inline SgfRecord parse_sgf(const std::string& text) {
  using namespace detail;
  SgfRecord rec;
  const int end = (int)text.size();
  int i = 0;
  while (i < end && text[i] != ';') ++i;
  if (i >= end) return rec;  // no header node: "Failed to read the header" (sgf.cc:54)
  ++i;
  bool bad_number = false;
  i = scan_node(text, i, end, [&](const std::string& rk, const std::string& rv) {
    const std::string k = ref_trim(rk), v = ref_trim(rv);
    if (k == "RE") {
      if (!v.empty()) rec.winner = (v[0] == 'B' || v[0] == 'b') ? 1 : 2;
    } else if (k == "SZ") {
      if (!to_int(v, &rec.board_size)) bad_number = true;
    } else if (k == "KM") {
      if (!to_float(v, &rec.komi)) bad_number = true;
    } else if (k == "HA") {
      if (!to_int(v, &rec.handi)) bad_number = true;
    }
  });
  // the reference's stoi/stof would throw out of the game thread here; we reject the record
  if (bad_number) return rec;

  // move nodes
  while (true) {
    while (i < end && text[i] != ';') ++i;
    if (i >= end) break;
    ++i;
    int move = -1, player = 0;
    if (i < end && text[i] == '(') {
      // ";(" : the reference recurses into a child chain that swallows the rest of the file,
      // comes back with offset 0 and gives the node up as corrupted (sgf.cc:207-214): the
      // main chain ends at the previous node.
      break;
    } else {
      i = scan_node(text, i, end, [&](const std::string& rk, const std::string& rv) {
        if (rk.size() != 1) return;
        if (rk[0] == 'B') { player = 1; move = action_of(ref_trim(rv)); }
        else if (rk[0] == 'W') { player = 2; move = action_of(ref_trim(rv)); }
      });
    }
    rec.moves.push_back((int16_t)move);
    rec.players.push_back((uint8_t)player);
  }
  if (rec.moves.empty()) return rec;  // iter.done() right away: load_game returns false
  rec.num_moves = 0;
  for (int16_t m : rec.moves) {
    if (m < 0) break;
    ++rec.num_moves;
  }
  rec.ok = true;
  return rec;
}

inline SgfRecord load_sgf_file(const std::string& path) {
  std::ifstream f(path);
  std::stringstream ss;
  ss << f.rdbuf();
  return parse_sgf(ss.str());
}

}  // namespace gogame
