// mg_vis.h — bitmask form of gym-minigrid's Grid.process_vis, shared by the
// HIP kernels and the host-side test hook mg_debug_vis_masks (exhaustively compared with the
// oracle's literal two-pass loops in tests/test_vis_bitmask.py).
//
// View coordinates: column i = vx (bit i of a row mask), row j = vy; the observer sits at
// (V/2, V-1) and rows are processed from j = V-1 up to j = 0 (called from
// envs/gym_multigrid/multigrid.py:1020-1022 with agent_pos = (V//2, V-1)).
//
// Per row: the incoming mask floods left and right through transparent cells (a visible
// transparent cell makes both horizontal neighbours visible); every visible transparent cell
// then makes the three cells above it visible.
//
// The flood is evaluated in closed form with a carry chain instead of V-1 shift-and-or rounds: with
// t = transparent cells of the row and s = m & t the visible transparent ones, t + s carries from
// each seed through the rest of its run of ones and leaves a one in the cell that ends the run, so
// (t + s) ^ t = every cell from the lowest seed of a run up to and including the first cell past the
// run, except seeds above the lowest one (their bit is 1 + 1 + carry = 1, unchanged) — and those are
// in m already.  The flood towards lower columns is the same on the bit-reversed row.
#pragma once
#include <stdint.h>

#define MG_HD __host__ __device__ __forceinline__   // only ever compiled by hipcc (mgrid.hip)

// opq[j] = opacity bits of row j (bit i set <=> cell (i,j) is opaque).  vis[j] out.
template <int V>
MG_HD void mg_vis_rows(const uint32_t (&opq)[V], uint32_t (&vis)[V]) {
  constexpr uint32_t full = (1u << V) - 1u;
#pragma unroll
  for (int j = 0; j < V; ++j) vis[j] = 0;
  vis[V - 1] = 1u << (V / 2);
#pragma unroll
  for (int j = V - 1; j >= 0; --j) {
    const uint32_t t = ~opq[j] & full;
    const uint32_t s = vis[j] & t;
    const uint32_t tr = __builtin_bitreverse32(t), sr = __builtin_bitreverse32(s);
    const uint32_t up = (t + s) ^ t, down = __builtin_bitreverse32((tr + sr) ^ tr);
    const uint32_t m = (vis[j] | up | down) & full;
    vis[j] = m;
    if (j > 0) {
      const uint32_t T = m & t;
      vis[j - 1] |= (T | (T << 1) | (T >> 1)) & full;
    }
  }
}
