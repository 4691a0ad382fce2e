// mgrid.hip — MI355X (gfx950) kernels and C ABI of the batched MultiGrid step path.
//
// One lane owns one env; one wavefront owns 64 consecutive envs.  Per launch every env does
//   step_count++ -> agent order -> per-agent action resolution (serial in that order) ->
//   respawn -> collective done -> v8 reward shaping -> optional auto-reset -> egocentric obs
// i.e. MultiGridEnv.step (envs/gym_multigrid/multigrid.py:963-995) + gen_obs (:1037-1083) +
// v8 shape_rewards (v8_robust_ppo.py:171-220), for N envs at once.
//
// Memory plan (all arrays agent-major, see include/mgrid.h):
//   reads : actions A B, agent words 4A B, shaper words 4A B, step word 4 B      (coalesced dwords)
//   writes: obs image A*V*V*3 B (staged in LDS, stored as whole 16-B vectors of a contiguous
//           64-env chunk), obs dir A B, reward 8A B, done 1 B, state write-back
//   LDS   : padded static map (one dword per cell), goal-distance LUT, 64*V*V*3 B staging per wave
// No inter-workgroup communication, no atomics except the (rare) error word.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>

#include "mgrid.h"
#include "mg_vis.h"

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kWave = 64;
constexpr int kMaxAttempts = 256;    // rejection-sampling cap, then a deterministic scan
enum { P_ORDER = 0, P_RESPAWN = 1, P_SPAWN = 2, P_ACTION = 3 };
enum { OP_STEP = 0, OP_RESET = 1 };

// map cell word: type | color<<8 | state<<16 | flags<<24
constexpr uint32_t F_OPAQUE = 1u << 24;  // !see_behind()
constexpr uint32_t F_FREE = 1u << 25;    // grid.get(...) is None
constexpr uint32_t F_KILL = 1u << 26;    // goal or lava: agent_is_done
constexpr uint32_t F_GOAL = 1u << 27;

struct Params {
  int32_t W, H, max_steps, see_through, competitive, shaping, auto_reset, act_elem;
  int32_t op, random_actions, PW, PH;
  int64_t n_envs, stride;
  uint64_t gid_base, t;
  uint32_t seed_lo, seed_hi;
  const uint32_t* map_pad;     // [PH][PW]
  const double* dist_lut;      // [H][W] distance to the shaper's goal cell
  const double* reward_lut;    // [max_steps+1]
  const void* actions;
  const uint8_t* order;
  const uint32_t* respawn;     // int8x4 per (agent, env)
  const uint32_t* reset_spawn; // int8x4 per (agent, env)
  const uint8_t* mask;
  uint8_t* actions_out;
  uint8_t* obs_image;
  uint8_t* obs_dir;
  double* reward;
  double* env_reward;
  uint8_t* done;
  uint32_t* agent_state;
  uint32_t* shaper_state;
  uint32_t* step_state;
  uint32_t* err;
};

struct U4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al. SC'11).  Counter: (global env id, t lo, t hi,
// purpose<<28 | agent<<20 | attempt); key = seed.  Same definition as oracle/mg_oracle.c.
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

template <int A>
__device__ __forceinline__ uint32_t sel(const uint32_t (&v)[A], int a) {
  uint32_t r = v[0];
#pragma unroll
  for (int i = 1; i < A; ++i) r = (a == i) ? v[i] : r;
  return r;
}
template <int A>
__device__ __forceinline__ void put(uint32_t (&v)[A], int a, uint32_t val) {
#pragma unroll
  for (int i = 0; i < A; ++i) v[i] = (a == i) ? val : v[i];
}

// place_obj's acceptance test (multigrid.py:603-613) on the shared static map.
// occ = bit mask of agents whose position currently counts (agent_pos not None).
template <int A>
__device__ __forceinline__ bool cell_free(const Params& p, const uint32_t* map, const uint32_t (&ag)[A],
                                          uint32_t occ, int x, int y) {
  if ((unsigned)x >= (unsigned)p.W || (unsigned)y >= (unsigned)p.H) return false;
  const int pad = (p.PW - p.W) >> 1;
  if (!(map[(y + pad) * p.PW + x + pad] & F_FREE)) return false;
  const uint32_t pos = (uint32_t)x | ((uint32_t)y << 8);
  bool ok = true;
#pragma unroll
  for (int b = 0; b < A; ++b) ok = ok && !(((occ >> b) & 1u) && (ag[b] & 0xFFFFu) == pos);
  return ok;
}

// place_one_agent + place_agent_at_pos (multigrid.py:646-675): returns x | y<<8 | dir<<16.
// inj != nullptr: the draw the reference accepted is injected (int8x4 = x, y, dir, 0).
template <int A>
__device__ __forceinline__ uint32_t place_agent(const Params& p, const uint32_t* map, const uint32_t (&ag)[A], uint32_t occ,
                                int a, int purpose, const uint32_t* inj, int64_t n, uint32_t gid, uint32_t& err) {
  int x = -1, y = -1, dir = 0;
  if (inj) {
    const uint32_t w = inj[(int64_t)a * p.stride + n];
    x = (int8_t)(w & 255u); y = (int8_t)((w >> 8) & 255u); dir = (int8_t)((w >> 16) & 255u);
    if (x < 0) { err |= MG_ERR_MISSING_DRAW; x = -1; }
    else if (!cell_free<A>(p, map, ag, occ, x, y) || (unsigned)dir > 3u) { err |= MG_ERR_BAD_RESPAWN; x = -1; }
  } else {
    for (int k = 0; k < kMaxAttempts; ++k) {   // while True: multigrid.py:589-619 (bounded here)
      const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32),
                          ((uint32_t)purpose << 28) | ((uint32_t)a << 20) | (uint32_t)k, p.seed_lo, p.seed_hi);
      const int cx = (int)__umulhi(r.x, (uint32_t)p.W), cy = (int)__umulhi(r.y, (uint32_t)p.H);
      if (cell_free<A>(p, map, ag, occ, cx, cy)) { x = cx; y = cy; dir = (int)(r.z >> 30); break; }
    }
  }
  if (x < 0) {   // exit condition of the sampling loop: first free cell in storage order, dir 0
    dir = 0;
    for (int j = 0; j < p.H && x < 0; ++j)
      for (int i = 0; i < p.W; ++i)
        if (cell_free<A>(p, map, ag, occ, i, j)) { x = i; y = j; break; }
    if (x < 0) { x = 0; y = 0; }
  }
  return (uint32_t)x | ((uint32_t)y << 8) | ((uint32_t)dir << 16);
}

// Egocentric view of agent `a` (gen_obs_grid + encode, multigrid.py:997-1035, 1081) for the 64
// envs of this wave, staged in LDS and stored to HBM as one contiguous 64*V*V*3-byte chunk.
//
// Closed form of slice + rotate_left x (dir+1): view cell (vx,vy) is world cell
//   pos + f*(V-1-vy) + r*(vx - V/2),  f = DIR_TO_VEC[dir], r = (-f.y, f.x)
// (equivalent to get_view_coords, multigrid.py:723-741).  Another agent seen in the view is
// encoded (10, id, (dir_other - dir_obs - 1) mod 4) (rotate_left's dir-1 per turn, :289-295).
template <int A, int V>
__device__ __forceinline__ void write_obs(const Params& p, const uint32_t* map, const uint32_t (&ag)[A], int a,
                                          uint32_t* stage, int lane, uint8_t* gchunk) {
  constexpr int ROWB = V * V * 3;           // bytes of one image
  constexpr int NW = (ROWB + 3) / 4;        // dwords of one image (last one holds 3 bytes: ROWB % 4 == 3 for odd V)
  static_assert((ROWB & 3) == 3, "odd view sizes only");
  const int pad = (p.PW - p.W) >> 1;
  const uint32_t me = ag[a];
  const int x = (int)(me & 255u), y = (int)((me >> 8) & 255u), d = (int)((me >> 16) & 3u);
  const int fx = (d == 0) - (d == 2), fy = (d == 1) - (d == 3);
  const int rx = -fy, ry = fx;
  const int dr = rx + p.PW * ry, df = fx + p.PW * fy;
  const int i0 = (y + pad + fy * (V - 1) - ry * (V / 2)) * p.PW + (x + pad + fx * (V - 1) - rx * (V / 2));

  uint32_t code[V * V];
  uint32_t opq[V], vis[V];
#pragma unroll
  for (int j = 0; j < V; ++j) opq[j] = 0;
#pragma unroll
  for (int vx = 0; vx < V; ++vx)
#pragma unroll
    for (int vy = 0; vy < V; ++vy) {
      const uint32_t w = map[i0 + vx * dr - vy * df];
      code[vx * V + vy] = w & 0xFFFFFFu;
      opq[vy] |= ((w >> 24) & 1u) << vx;
    }
  if (p.see_through) {
#pragma unroll
    for (int j = 0; j < V; ++j) vis[j] = (1u << V) - 1u;
  } else {
    mg_vis_rows<V>(opq, vis);
  }
  // own cell <- carried object or None, after process_vis (multigrid.py:1029-1033)
  {
    const uint32_t carry = me >> 24;
    code[(V / 2) * V + (V - 1)] = carry ? ((carry >> 3) | ((carry & 7u) << 8)) : (uint32_t)MG_T_EMPTY;
  }

  // pack the V*V 3-byte codes into a dword stream (all positions compile-time)
  uint32_t s[NW + 1];
#pragma unroll
  for (int i = 0; i <= NW; ++i) s[i] = 0;
#pragma unroll
  for (int vx = 0; vx < V; ++vx)
#pragma unroll
    for (int vy = 0; vy < V; ++vy) {
      const int c = vx * V + vy;
      const uint32_t v = ((vis[vy] >> vx) & 1u) ? code[c] : 0u;   // unseen -> (0,0,0)
      const int b = 3 * c, wi = b >> 2, sh = (b & 3) * 8;
      s[wi] |= v << sh;
      if (sh > 8) s[wi + 1] |= v >> (32 - sh);
    }

  // This lane's image occupies bytes [lane*ROWB, lane*ROWB+ROWB) of the wave chunk, which is not
  // dword aligned.  A lane owns the chunk dwords that START inside its byte range; the last of
  // them is completed with the first bytes of the next lane's stream.
  const uint32_t nxt0 = __shfl_down(s[0], 1);
  s[NW - 1] |= (nxt0 & 255u) << 24;
  s[NW] = nxt0 >> 8;
  const int b0 = lane * ROWB;
  const uint32_t h = (4u - ((uint32_t)b0 & 3u)) & 3u;   // stream bytes owned by the previous lane's last dword
  const int d0 = (b0 + 3) >> 2;
  const int cnt = (ROWB - (int)h + 3) >> 2;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const uint32_t v = __builtin_amdgcn_alignbyte(s[j + 1], s[j], h);
    if (j < NW - 1 || j < cnt) stage[d0 + j] = v;
  }
  // other agents: patch their cell (3 bytes) if it lies in the view and is visible
  uint8_t* sb = reinterpret_cast<uint8_t*>(stage) + b0;
#pragma unroll
  for (int b = 0; b < A; ++b) {
    if (b == a) continue;
    const uint32_t o = ag[b];
    const int ox = (int)(o & 255u) - x, oy = (int)((o >> 8) & 255u) - y;
    const int along = ox * fx + oy * fy, side = ox * rx + oy * ry;
    const int vy = V - 1 - along, vx = side + V / 2;
    if ((unsigned)vx < (unsigned)V && (unsigned)vy < (unsigned)V) {
      uint32_t vrow = vis[0];
#pragma unroll
      for (int j = 1; j < V; ++j) vrow = (vy == j) ? vis[j] : vrow;
      if ((vrow >> vx) & 1u) {
        uint8_t* q = sb + (vx * V + vy) * 3;
        q[0] = (uint8_t)MG_T_AGENT;
        q[1] = (uint8_t)b;
        q[2] = (uint8_t)((((o >> 16) & 3u) - (uint32_t)d - 1u) & 3u);
      }
    }
  }
  // LDS is processed in order per wave, so the reads below see the writes above; the compiler
  // only needs to be kept from reordering them (cross-lane dependence it cannot see).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int NV = 4 * ROWB;                // 16-byte vectors in the 64-image chunk
  const uint4* src = reinterpret_cast<const uint4*>(stage);
  uint4* dst = reinterpret_cast<uint4*>(gchunk);
#pragma unroll
  for (int i = 0; i < (NV + kWave - 1) / kWave; ++i) {
    const int idx = i * kWave + lane;
    if (idx < NV) dst[idx] = src[idx];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int A, int V>
__global__ __launch_bounds__(kBlock) void mg_kernel(const Params p) {
  constexpr int ROWB = V * V * 3;
  extern __shared__ __align__(16) uint8_t smem[];
  // LDS carve: [stage: 4 waves x 64*ROWB B][dist LUT: W*H f64][map: PW*PH u32]
  uint32_t* stage_all = reinterpret_cast<uint32_t*>(smem);
  double* dist = reinterpret_cast<double*>(smem + (kBlock / kWave) * kWave * ROWB);
  uint32_t* map = reinterpret_cast<uint32_t*>(dist + p.W * p.H);

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  for (int i = tid; i < p.PW * p.PH; i += kBlock) map[i] = p.map_pad[i];
  for (int i = tid; i < p.W * p.H; i += kBlock) dist[i] = p.dist_lut[i];
  __syncthreads();

  const int64_t env0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wv) * kWave;
  if (env0 >= p.stride) return;                 // whole wave beyond the allocation (wave-uniform)
  const int64_t n = env0 + lane;
  const bool active = n < p.n_envs;
  const uint32_t gid = (uint32_t)(p.gid_base + (uint64_t)n);
  uint32_t* stage = stage_all + wv * (kWave * ROWB / 4);

  uint32_t ag[A], sh[A], acts[A], st, err = 0;
  double rew[A];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    ag[a] = p.agent_state[(int64_t)a * p.stride + n];
    sh[a] = p.shaper_state[(int64_t)a * p.stride + n];
    // keep every later LDS index in range whatever the buffers hold
    const uint32_t cx = min(ag[a] & 255u, (uint32_t)(p.W - 1)), cy = min((ag[a] >> 8) & 255u, (uint32_t)(p.H - 1));
    ag[a] = (ag[a] & 0xFF030000u) | cx | (cy << 8);
    rew[a] = 0.0;
    acts[a] = 6;
  }
  st = p.step_state[n];
  uint32_t done_flag = 0;
  const int pad = (p.PW - p.W) >> 1;

  if (p.op == OP_STEP) {
    // ---- actions -------------------------------------------------------------------------
    if (p.random_actions) {
      const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32), (uint32_t)P_ACTION << 28, p.seed_lo, p.seed_hi);
      const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int a = 0; a < A; ++a) acts[a] = __umulhi(rr[a & 3], 7u);
      if (p.actions_out && active) {
#pragma unroll
        for (int a = 0; a < A; ++a) p.actions_out[(int64_t)a * p.stride + n] = (uint8_t)acts[a];
      }
    } else {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const int64_t i = (int64_t)a * p.stride + n;
        if (p.act_elem == MG_ACT_U8) acts[a] = static_cast<const uint8_t*>(p.actions)[i];
        else if (p.act_elem == MG_ACT_I32) acts[a] = (uint32_t)static_cast<const int32_t*>(p.actions)[i];
        else acts[a] = (uint32_t)static_cast<const int64_t*>(p.actions)[i];
      }
    }
    // ---- step_count += 1 (multigrid.py:968) ----------------------------------------------
    uint32_t step = (st & 0xFFFFu) + 1u;
    uint32_t dbits = st >> 16;
    if (step > 0xFFFFu) step = 0xFFFFu;

    // ---- agent order (multigrid.py:973-974) ----------------------------------------------
    int ord[A];
#pragma unroll
    for (int k = 0; k < A; ++k) ord[k] = k;
    if (p.order) {
      uint32_t seen = 0;
#pragma unroll
      for (int k = 0; k < A; ++k) {
        ord[k] = p.order[(int64_t)k * p.stride + n];
        if (ord[k] < A) seen |= 1u << ord[k];
      }
      if (seen != (1u << A) - 1u) {
        if (active) err |= MG_ERR_BAD_ORDER;
#pragma unroll
        for (int k = 0; k < A; ++k) ord[k] = k;
      }
    } else if (A > 1) {
      const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32), (uint32_t)P_ORDER << 28, p.seed_lo, p.seed_hi);
      const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
      // Fisher-Yates from the top, as np.random.shuffle does; static swaps via selects
#pragma unroll
      for (int i = A - 1, w = 0; i >= 1; --i, ++w) {
        const int j = (int)__umulhi(rr[w & 3], (uint32_t)(i + 1));
        const int oi = ord[i];
        int oj = ord[0];
#pragma unroll
        for (int q = 1; q < A; ++q) oj = (j == q) ? ord[q] : oj;
#pragma unroll
        for (int q = 0; q < A; ++q) ord[q] = (j == q) ? oi : ord[q];
        ord[i] = oj;
      }
    }

    // ---- step_one_agent in that order (multigrid.py:860-924) ------------------------------
#pragma unroll
    for (int k = 0; k < A; ++k) {
      const int a = ord[k];
      uint32_t s = sel<A>(ag, a);
      const uint32_t act = sel<A>(acts, a);
      int x = (int)(s & 255u), y = (int)((s >> 8) & 255u), d = (int)((s >> 16) & 3u);
      const uint32_t hi = s & 0xFF000000u;
      const int fx = x + (d == 0) - (d == 2), fy = y + (d == 1) - (d == 3);   // front_pos, pre-action
      double r = 0.0;
      if (act == 0u) {
        d = (d + 3) & 3;
      } else if (act == 1u) {
        d = (d + 1) & 3;
      } else if (act == 2u) {
        const uint32_t fpos = (uint32_t)fx | ((uint32_t)fy << 8);
        bool blocked = false;
#pragma unroll
        for (int b = 0; b < A; ++b) blocked = blocked || (b != a && (ag[b] & 0xFFFFu) == fpos);
        if (!blocked) {
          const uint32_t cell = map[(fy + pad) * p.PW + fx + pad];
          if (cell & F_KILL) {                       // goal or lava: agent_is_done (:815-832)
            dbits |= 1u << a;
            if (cell & F_GOAL)                       // _reward(), re-read cell is still the goal (:881-883)
              r = (step <= (uint32_t)p.max_steps) ? p.reward_lut[step]
                                                  : 1.0 - 0.9 * ((double)step / (double)p.max_steps);
            const uint32_t np_ = place_agent<A>(p, map, ag, ((1u << A) - 1u) & ~(1u << a), a, P_RESPAWN,
                                                p.respawn, n, gid, err);
            x = (int)(np_ & 255u); y = (int)((np_ >> 8) & 255u); d = (int)((np_ >> 16) & 3u);
          } else if (cell & F_FREE) {                // move_agent (:834-847)
            x = fx; y = fy;
          }                                           // wall: nothing moves, still "successful"
        }
      } else if (act > 6u) {
        err |= MG_ERR_BAD_ACTION;                     // assert False (:902)
      }                                               // pickup/drop/toggle/done: no-ops on a static map
      s = (uint32_t)x | ((uint32_t)y << 8) | ((uint32_t)d << 16) | hi;
      put<A>(ag, a, s);
#pragma unroll
      for (int i = 0; i < A; ++i) rew[i] = (a == i) ? r : rew[i];
    }

    // ---- collective done (multigrid.py:986-993) -------------------------------------------
    done_flag = (p.competitive && dbits != 0u) || (step >= (uint32_t)p.max_steps);

    // ---- v8 shape_rewards (v8_robust_ppo.py:171-220), float64, one rounding per operation ---
    double out_rew[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      out_rew[a] = rew[a];
      if (p.shaping == MG_SHAPING_V8) {
        const int x = (int)(ag[a] & 255u), y = (int)((ag[a] >> 8) & 255u);
        uint32_t w = sh[a];
        if (rew[a] > 0.0) {                           // v8:181-186
          out_rew[a] = 5.0;
          w = 0u;
        } else {
          double r = 0.0;
          const bool has = (w >> 16) & 1u;
          const int px = (int)(w & 255u), py = (int)((w >> 8) & 255u);
          uint32_t stat = w >> 17;
          const double cur = dist[y * p.W + x];       // np.linalg.norm(pos - goal), v8:192
          if (has) {
            const double prev = dist[min(py, p.H - 1) * p.W + min(px, p.W - 1)];
            r = r + (prev - cur) * 0.2;               // v8:193-195
            if (px == x && py == y) {                 // v8:199-206
              stat += 1u;
              if (stat > 3u) r = r - 0.05;
            } else {
              stat = 0u;
              r = r + 0.02;
            }
          }
          if (acts[a] == 2u) r = r + 0.02;            // v8:209-212
          else if (acts[a] <= 1u) r = r + 0.01;
          if (-0.2 > r) r = -0.2;                     // max(reward, -0.2), v8:215
          out_rew[a] = r;
          if (stat > 0x7FFFu) stat = 0x7FFFu;
          w = (uint32_t)x | ((uint32_t)y << 8) | (1u << 16) | (stat << 17);   // v8:218
        }
        sh[a] = w;
      }
    }

    // ---- auto-reset: reset() of the finished env inside the same launch --------------------
    st = step | (dbits << 16);
    if (done_flag && p.auto_reset) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const uint32_t np_ = place_agent<A>(p, map, ag, (1u << a) - 1u, a, P_SPAWN, p.reset_spawn, n, gid, err);
        ag[a] = np_;                                  // carrying cleared (multigrid.py:490)
        sh[a] = 0u;                                   // v8:227-229
      }
      st = 0u;
    }

    if (active) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const int64_t i = (int64_t)a * p.stride + n;
        p.reward[i] = out_rew[a];
        if (p.env_reward) p.env_reward[i] = rew[a];
      }
      p.done[n] = (uint8_t)done_flag;
    }
  } else {
    // ---- reset() (multigrid.py:466-501; cluttered.py:55-63) --------------------------------
    const bool m = p.mask ? (p.mask[active ? n : 0] != 0) : true;
    if (m) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const uint32_t np_ = place_agent<A>(p, map, ag, (1u << a) - 1u, a, P_SPAWN, p.reset_spawn, n, gid, err);
        ag[a] = np_;
        sh[a] = 0u;
      }
      st = 0u;
    }
  }

  if (active) {
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const int64_t i = (int64_t)a * p.stride + n;
      p.agent_state[i] = ag[a];
      p.shaper_state[i] = sh[a];
      p.obs_dir[i] = (uint8_t)((ag[a] >> 16) & 3u);   // obs['direction'][a] (multigrid.py:1083)
    }
    p.step_state[n] = st;
    if (err) atomicOr(p.err, err);
  }

  // ---- gen_obs (multigrid.py:1037-1067) -----------------------------------------------------
#pragma unroll
  for (int a = 0; a < A; ++a)
    write_obs<A, V>(p, map, ag, a, stage, lane, p.obs_image + ((int64_t)a * p.stride + env0) * ROWB);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(MG_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
  } while (0)

typedef void (*kernel_fn)(const Params);

struct Variant { int A, V; kernel_fn fn; };
const Variant kVariants[] = {
    {1, 5, mg_kernel<1, 5>}, {2, 5, mg_kernel<2, 5>}, {3, 5, mg_kernel<3, 5>}, {4, 5, mg_kernel<4, 5>},
    {1, 7, mg_kernel<1, 7>}, {2, 7, mg_kernel<2, 7>}, {3, 7, mg_kernel<3, 7>}, {3, 3, mg_kernel<3, 3>},
};

}  // namespace

struct mg_handle {
  mg_config cfg;
  int device;
  hipStream_t stream;
  bool own_stream;
  kernel_fn fn;
  int64_t stride;
  int PW, PH;
  size_t lds_bytes;
  bool have_map, have_reset;
  uint64_t t;
  int gx, gy;
  // device memory
  uint32_t* map_pad;
  double* dist_lut;
  double* reward_lut;
  uint32_t* err;
  mg_buffers buf;
  std::vector<void*> allocs;
};

namespace {

struct DeviceGuard {
  int prev;
  bool ok;
  explicit DeviceGuard(int dev) : prev(0), ok(false) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() { if (ok) (void)hipSetDevice(prev); }
};

template <typename T>
int dalloc(mg_handle* h, T** out, size_t count) {
  void* ptr = nullptr;
  hipError_t e = hipMalloc(&ptr, count * sizeof(T));
  if (e != hipSuccess) return fail(MG_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  h->allocs.push_back(ptr);
  e = hipMemset(ptr, 0, count * sizeof(T));
  if (e != hipSuccess) return fail(MG_EHIP, std::string("hipMemset: ") + hipGetErrorString(e));
  *out = static_cast<T*>(ptr);
  return MG_OK;
}

int launch(mg_handle* h, int op, const void* actions, const uint8_t* order, const int8_t* respawn,
           const int8_t* reset_spawn, const uint8_t* mask, int random_actions, uint8_t* actions_out) {
  Params p;
  memset(&p, 0, sizeof(p));
  const mg_config& c = h->cfg;
  p.W = c.width; p.H = c.height; p.max_steps = c.max_steps; p.see_through = c.see_through_walls;
  p.competitive = (c.n_agents == 1) ? 1 : c.competitive;   // multigrid.py:380-381
  p.shaping = c.shaping_kind; p.auto_reset = c.auto_reset; p.act_elem = c.action_dtype;
  p.op = op; p.random_actions = random_actions; p.PW = h->PW; p.PH = h->PH;
  p.n_envs = c.n_envs; p.stride = h->stride; p.gid_base = (uint64_t)c.env_id_base; p.t = h->t;
  p.seed_lo = (uint32_t)c.seed; p.seed_hi = (uint32_t)(c.seed >> 32);
  p.map_pad = h->map_pad; p.dist_lut = h->dist_lut; p.reward_lut = h->reward_lut;
  p.actions = actions; p.order = order;
  p.respawn = reinterpret_cast<const uint32_t*>(respawn);
  p.reset_spawn = reinterpret_cast<const uint32_t*>(reset_spawn);
  p.mask = mask; p.actions_out = actions_out;
  p.obs_image = h->buf.obs_image; p.obs_dir = h->buf.obs_dir; p.reward = h->buf.reward;
  p.env_reward = h->buf.env_reward; p.done = h->buf.done; p.agent_state = h->buf.agent_state;
  p.shaper_state = h->buf.shaper_state; p.step_state = h->buf.step_state; p.err = h->err;
  const unsigned grid = (unsigned)((h->stride + kBlock - 1) / kBlock);
  DeviceGuard g(h->device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  hipLaunchKernelGGL(h->fn, dim3(grid), dim3(kBlock), h->lds_bytes, h->stream, p);
  HIP_TRY(hipGetLastError());
  return MG_OK;
}

template <int V>
void vis_hook(const uint64_t* patterns, int64_t n, uint64_t* out) {
  for (int64_t k = 0; k < n; ++k) {
    const uint64_t pat = patterns[k];
    uint64_t res = 0;
    uint32_t opq[V], vis[V];
    for (int j = 0; j < V; ++j) opq[j] = 0;
    for (int i = 0; i < V; ++i)
      for (int j = 0; j < V; ++j) opq[j] |= (uint32_t)((pat >> (i * V + j)) & 1u) << i;
    mg_vis_rows<V>(opq, vis);
    for (int i = 0; i < V; ++i)
      for (int j = 0; j < V; ++j) res |= (uint64_t)((vis[j] >> i) & 1u) << (i * V + j);
    out[k] = res;
  }
}

}  // namespace

extern "C" {

const char* mg_last_error(void) { return g_err.c_str(); }
const char* mg_version(void) { return "mgrid-hip 0.1 (gfx950)"; }

int mg_create(const mg_config* cfg, int device, mg_handle** out) {
  if (!cfg || !out) return fail(MG_EINVAL, "null argument");
  *out = nullptr;
  const mg_config& c = *cfg;
  if (c.width < 3 || c.height < 3 || c.width > 64 || c.height > 64) return fail(MG_EINVAL, "width/height must be in [3,64]");
  if (c.n_envs < 1) return fail(MG_EINVAL, "n_envs must be >= 1");
  if (c.max_steps < 1 || c.max_steps > 65534) return fail(MG_EINVAL, "max_steps must be in [1,65534]");
  if (c.shaping_kind != MG_SHAPING_NONE && c.shaping_kind != MG_SHAPING_V8) return fail(MG_EINVAL, "unknown shaping_kind");
  if (c.action_dtype != MG_ACT_U8 && c.action_dtype != MG_ACT_I32 && c.action_dtype != MG_ACT_I64)
    return fail(MG_EINVAL, "unknown action_dtype");
  if (c.env_id_base < 0 || (uint64_t)c.env_id_base + (uint64_t)c.n_envs > 0xFFFFFFFFull)
    return fail(MG_EINVAL, "global env ids must fit 32 bits");
  kernel_fn fn = nullptr;
  for (const Variant& v : kVariants)
    if (v.A == c.n_agents && v.V == c.view_size) fn = v.fn;
  if (!fn) return fail(MG_EUNSUPPORTED, "no kernel instantiated for this (n_agents, view_size)");

  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MG_EINVAL, "bad device index");
  DeviceGuard g(device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");

  mg_handle* h = new mg_handle();
  h->cfg = c; h->device = device; h->fn = fn; h->t = 0; h->have_map = h->have_reset = false;
  h->stride = (c.n_envs + 63) / 64 * 64;
  h->PW = c.width + 2 * (c.view_size - 1);
  h->PH = c.height + 2 * (c.view_size - 1);
  h->gx = h->gy = 13;
  const int rowb = c.view_size * c.view_size * 3;
  h->lds_bytes = (size_t)(kBlock / kWave) * kWave * rowb + (size_t)c.width * c.height * 8 + (size_t)h->PW * h->PH * 4;
  h->own_stream = true;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return fail(MG_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
  if (e != hipSuccess) { (void)hipStreamDestroy(h->stream); delete h; return fail(MG_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }

  const size_t A = (size_t)c.n_agents, S = (size_t)h->stride;
  memset(&h->buf, 0, sizeof(h->buf));
  int rc = MG_OK;
  if (rc == MG_OK) rc = dalloc(h, &h->map_pad, (size_t)h->PW * h->PH);
  if (rc == MG_OK) rc = dalloc(h, &h->dist_lut, (size_t)c.width * c.height);
  if (rc == MG_OK) rc = dalloc(h, &h->reward_lut, (size_t)c.max_steps + 1);
  if (rc == MG_OK) rc = dalloc(h, &h->err, 1);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.obs_image, A * S * rowb);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.obs_dir, A * S);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.reward, A * S);
  if (rc == MG_OK && c.emit_env_reward) rc = dalloc(h, &h->buf.env_reward, A * S);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.done, S);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.agent_state, A * S);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.shaper_state, A * S);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.step_state, S);
  if (rc == MG_OK) {
    // MiniGridEnv._reward() for every step count, computed with the host's float64 ops so that
    // the device returns exactly what CPython computes (multigrid.py:883)
    std::vector<double> lut((size_t)c.max_steps + 1);
    for (int s = 0; s <= c.max_steps; ++s) {
      volatile double q = (double)s / (double)c.max_steps;
      volatile double m = 0.9 * q;
      lut[(size_t)s] = 1.0 - m;
    }
    e = hipMemcpy(h->reward_lut, lut.data(), lut.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(MG_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  }
  if (rc != MG_OK) {
    for (void* q : h->allocs) (void)hipFree(q);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return rc;
  }
  h->buf.n_envs = c.n_envs; h->buf.env_stride = h->stride; h->buf.n_agents = c.n_agents; h->buf.obs_cell_bytes = rowb;
  *out = h;
  return MG_OK;
}

int mg_destroy(mg_handle* h) {
  if (!h) return MG_OK;
  DeviceGuard g(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (void* q : h->allocs) (void)hipFree(q);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MG_OK;
}

int mg_set_static_map(mg_handle* h, const uint8_t* cells) {
  if (!h || !cells) return fail(MG_EINVAL, "null argument");
  const int W = h->cfg.width, H = h->cfg.height, pad = h->cfg.view_size - 1;
  const uint32_t wall = (uint32_t)MG_T_WALL | (5u << 8) | F_OPAQUE;   // Grid.slice's out-of-bounds Wall() (multigrid.py:315)
  std::vector<uint32_t> m((size_t)h->PW * h->PH, wall);
  int gx = -1, gy = -1;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const uint8_t* c = cells + ((size_t)y * W + x) * 3;
      uint32_t w = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
      switch (c[0]) {
        case MG_T_EMPTY: w = (uint32_t)MG_T_EMPTY | F_FREE; break;
        case MG_T_WALL: w |= F_OPAQUE; break;
        case MG_T_GOAL: w |= F_KILL | F_GOAL; break;
        case MG_T_LAVA: w |= F_KILL; break;
        default:
          return fail(MG_EUNSUPPORTED, "static maps hold only empty/wall/goal/lava cells");
      }
      m[(size_t)(y + pad) * h->PW + (x + pad)] = w;
    }
  // get_goal_position (v8_robust_ppo.py:159-169): first goal scanning x-major, else (13,13)
  for (int i = 0; i < W && gx < 0; ++i)
    for (int j = 0; j < H; ++j)
      if (cells[((size_t)j * W + i) * 3] == MG_T_GOAL) { gx = i; gy = j; break; }
  if (gx < 0) { gx = 13; gy = 13; }
  h->gx = gx; h->gy = gy;
  std::vector<double> d((size_t)W * H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const long dx = x - gx, dy = y - gy;
      d[(size_t)y * W + x] = sqrt((double)(dx * dx + dy * dy));   // np.linalg.norm (v8:192)
    }
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->map_pad, m.data(), m.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dist_lut, d.data(), d.size() * 8, hipMemcpyHostToDevice));
  h->have_map = true;
  return MG_OK;
}

int mg_reset(mg_handle* h, const uint8_t* env_mask) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (!h->have_map) return fail(MG_ESTATE, "mg_set_static_map must be called before mg_reset");
  int rc = launch(h, OP_RESET, nullptr, nullptr, nullptr, nullptr, env_mask, 0, nullptr);
  if (rc == MG_OK && !env_mask) h->have_reset = true;
  if (rc == MG_OK && env_mask) h->have_reset = true;
  return rc;
}

int mg_reset_inject(mg_handle* h, const uint8_t* env_mask, const int8_t* spawn) {
  if (!h || !spawn) return fail(MG_EINVAL, "null argument");
  if (!h->have_map) return fail(MG_ESTATE, "mg_set_static_map must be called before mg_reset_inject");
  int rc = launch(h, OP_RESET, nullptr, nullptr, nullptr, spawn, env_mask, 0, nullptr);
  if (rc == MG_OK) h->have_reset = true;
  return rc;
}

int mg_step(mg_handle* h, const void* actions) {
  if (!h || !actions) return fail(MG_EINVAL, "null argument");
  if (!h->have_reset) return fail(MG_ESTATE, "mg_reset must be called before mg_step");
  int rc = launch(h, OP_STEP, actions, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
  if (rc == MG_OK) h->t += 1;
  return rc;
}

int mg_step_replay(mg_handle* h, const void* actions, const uint8_t* order, const int8_t* respawn,
                   const int8_t* reset_spawn) {
  if (!h || !actions || !order || !respawn) return fail(MG_EINVAL, "null argument");
  if (h->cfg.auto_reset && !reset_spawn) return fail(MG_EINVAL, "auto_reset replay needs reset_spawn");
  if (!h->have_reset) return fail(MG_ESTATE, "mg_reset must be called before mg_step_replay");
  int rc = launch(h, OP_STEP, actions, order, respawn, reset_spawn, nullptr, 0, nullptr);
  if (rc == MG_OK) h->t += 1;
  return rc;
}

int mg_step_random(mg_handle* h, uint8_t* actions_out) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (!h->have_reset) return fail(MG_ESTATE, "mg_reset must be called before mg_step_random");
  int rc = launch(h, OP_STEP, nullptr, nullptr, nullptr, nullptr, nullptr, 1, actions_out);
  if (rc == MG_OK) h->t += 1;
  return rc;
}

int mg_get_buffers(mg_handle* h, mg_buffers* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  *out = h->buf;
  return MG_OK;
}

int mg_set_stream(mg_handle* h, void* hip_stream) {
  if (!h) return fail(MG_EINVAL, "null handle");
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->own_stream) { (void)hipStreamDestroy(h->stream); h->own_stream = false; }
  h->stream = static_cast<hipStream_t>(hip_stream);
  return MG_OK;
}

int mg_get_stream(mg_handle* h, void** hip_stream) {
  if (!h || !hip_stream) return fail(MG_EINVAL, "null argument");
  *hip_stream = static_cast<void*>(h->stream);
  return MG_OK;
}

int mg_sync(mg_handle* h) {
  if (!h) return fail(MG_EINVAL, "null handle");
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return MG_OK;
}

int mg_get_step_index(mg_handle* h, uint64_t* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  *out = h->t;
  return MG_OK;
}

int mg_set_step_index(mg_handle* h, uint64_t t) {
  if (!h) return fail(MG_EINVAL, "null handle");
  h->t = t;
  return MG_OK;
}

int mg_check_errors(mg_handle* h, uint32_t* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out, h->err, 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(h->err, 0, 4));
  return MG_OK;
}

int mg_get_goal(mg_handle* h, int32_t* gx, int32_t* gy) {
  if (!h || !gx || !gy) return fail(MG_EINVAL, "null argument");
  *gx = h->gx; *gy = h->gy;
  return MG_OK;
}

// Test hook (host only, no GPU): the kernels' bitmask visibility on n opacity patterns.
int mg_debug_vis_masks(int V, const uint64_t* patterns, int64_t n, uint64_t* out) {
  if (!patterns || !out) return fail(MG_EINVAL, "null argument");
  switch (V) {
    case 3: vis_hook<3>(patterns, n, out); return MG_OK;
    case 5: vis_hook<5>(patterns, n, out); return MG_OK;
    case 7: vis_hook<7>(patterns, n, out); return MG_OK;
    default: return fail(MG_EINVAL, "V must be 3, 5 or 7");
  }
}

}  // extern "C"
