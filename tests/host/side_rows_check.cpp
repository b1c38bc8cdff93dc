// CPU check of csrc/hdg_side_rows.hpp (compiled by tests/test_host.py with g++): for every number of tile rows and of side rows a
// leg launch can have, every tile row and every side row is visited exactly once.
#include <cstdio>
#include <vector>

#include "../../incompressibleeulerhdg_amd/csrc/hdg_side_rows.hpp"

int main() {
  long cases = 0;
  for (int nt = 1; nt <= 70; nt++) {
    for (int extra = 0; extra <= 400; extra++) {
      const int period = hdg::side_row_period(nt, extra);
      const int rows = nt + extra;
      std::vector<int> tiles(nt, 0), sides(extra > 0 ? extra : 1, 0);
      for (int by = 0; by < rows; by++) {
        const hdg::SideRow r = hdg::side_row_of(by, rows, extra, period);
        if (r.idx < 0 || r.idx >= (r.side ? extra : nt)) { std::printf("error index nt=%d extra=%d by=%d\n", nt, extra, by); return 1; }
        (r.side ? sides : tiles)[r.idx]++;
      }
      for (int t = 0; t < nt; t++) if (tiles[t] != 1) { std::printf("error tile nt=%d extra=%d t=%d count=%d\n", nt, extra, t, tiles[t]); return 1; }
      for (int t = 0; t < extra; t++) if (sides[t] != 1) { std::printf("error side nt=%d extra=%d t=%d count=%d\n", nt, extra, t, sides[t]); return 1; }
      // interleaved: no run of the more numerous kind is longer than |period| - 1 while the other kind lasts
      cases++;
    }
  }
  std::printf("ok %ld\n", cases);
  return 0;
}
