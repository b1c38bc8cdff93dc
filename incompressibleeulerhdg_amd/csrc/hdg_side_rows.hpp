// Rows of a V-cycle leg launch that carries side jobs (hdg_kernels.hpp: SideXP, HDG_P1_SIDE_JOB; Engine::xp_side_slice).
// A leg launch over nt tile rows gets `extra` more rows of workgroups; each workgroup of such a SIDE row updates 1024 sixteen-byte
// pairs of the condensed CG's p and x instead of a tile.  The rows are interleaved: the less numerous kind of row (side rows for
// period > 0, tile rows for period < 0) takes the last row of every group of |period| rows until it is used up, so that the
// short-lived side workgroups are dispatched in between the tiles instead of behind them (the finest level has twice as many
// tiles as the chip has slots for).  Plain C++ (constexpr: usable on the device as it stands), so that the map -- every tile row
// and every side row exactly once, for every (nt, extra) -- is checked on the CPU: tests/host/side_rows_check.cpp.
#pragma once

namespace hdg {

struct SideRow {
  bool side;  // the workgroup row does a side job (idx = which of the `extra` side rows) or a tile row (idx = which of the nt)
  int idx;
};
// by: row of the workgroup in the launch, grid_rows = nt + extra
constexpr SideRow side_row_of(int by, int grid_rows, int extra, int period) {
  if (extra <= 0) return SideRow{false, by};
  const int per = period < 0 ? -period : period;
  const int nminor = period < 0 ? grid_rows - extra : extra;
  const int q = by / per, rem = by - q * per;
  const bool minor = rem == per - 1 && q < nminor;
  const int idx = minor ? q : by - (q < nminor ? q : nminor);
  return SideRow{(period < 0) != minor, idx};
}
// the interleaving period for nt tile rows and `extra` side rows (host)
constexpr int side_row_period(int nt, int extra) {
  if (extra <= 0) return 2;
  const int p = extra <= nt ? (nt + extra) / extra : (nt + extra) / nt;
  return extra <= nt ? (p < 2 ? 2 : p) : -(p < 2 ? 2 : p);
}

}  // namespace hdg
