// mgrid.hip — MI355X (gfx950) kernels and C ABI of the batched MultiGrid step path.
//
// One lane owns one env; one wavefront owns 64 consecutive envs ("chunk").  Per launch every env does
//   step_count++ -> agent order -> per-agent action resolution (serial in that order) ->
//   respawn -> collective done -> v8 reward shaping -> optional auto-reset -> egocentric obs
// i.e. MultiGridEnv.step (envs/gym_multigrid/multigrid.py:963-995) + gen_obs (:1037-1083) +
// v8 shape_rewards (v8_robust_ppo.py:171-220), for N envs at once.
//
// Memory plan (see include/mgrid.h for the layouts):
//   reads : actions A B, env_state 4(A+1) B (one 16-B vector per env for A=3)
//   writes: obs image A*V*V*3 B (staged in LDS, stored as whole 16-B vectors of a contiguous
//           64-env chunk), obs dir A B, reward 8A B, done 1 B, env_state write-back
//   LDS   : per-wave 64*V*V*3 B staging; padded static map (one dword per cell); goal-distance
//           LUT; and, on the table path, the precomputed static view of every (free cell, dir).
// Workgroups are persistent (grid-stride over chunks) so the LDS tables are filled once per CU.
// No inter-workgroup communication, no atomics except the (rare) error word.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "mgrid.h"
#include "mg_vis.h"

// Ablation switch for timing-only builds (tools/build_variant.sh): 0 = product,
// 1 = skip the obs global stores (LDS staging kept), 2 = skip obs entirely.
#ifndef MG_ABL
#define MG_ABL 0
#endif

namespace {

constexpr int kWave = 64;
constexpr int kMaxBlock = 1024;
constexpr int kMaxAttempts = 256;    // rejection-sampling cap, then a deterministic scan
constexpr int kLdsLimit = 160 * 1024;
constexpr int kUpfront = 4;         // chunks per wave whose inputs are all fetched before the first store
enum { P_ORDER = 0, P_RESPAWN = 1, P_SPAWN = 2, P_ACTION = 3 };
enum { OP_STEP = 0, OP_RESET = 1 };

// map cell word: type | color<<8 | state<<16 | flags<<24
constexpr uint32_t F_OPAQUE = 1u << 31;  // !see_behind(); the top bit, so one v_alignbit shifts it into a row mask (view_row_bit)
constexpr uint32_t F_FREE = 1u << 25;    // grid.get(...) is None
constexpr uint32_t F_KILL = 1u << 26;    // goal or lava: agent_is_done
constexpr uint32_t F_GOAL = 1u << 27;

// agent word of env_state (include/mgrid.h): x:6 | y:6 | dir:2 | prev_x:6 | prev_y:6 | valid:1 | stationary:3
constexpr uint32_t AW_POS = 0xFFFu;          // x | y<<6
constexpr uint32_t AW_POSDIR = 0x3FFFu;
__host__ __device__ __forceinline__ int aw_x(uint32_t w) { return (int)(w & 63u); }
__host__ __device__ __forceinline__ int aw_y(uint32_t w) { return (int)((w >> 6) & 63u); }
__host__ __device__ __forceinline__ int aw_d(uint32_t w) { return (int)((w >> 12) & 3u); }

template <int V> struct ViewGeom {
  static constexpr int ROWB = V * V * 3;            // bytes of one image
  static constexpr int NW = (ROWB + 3) / 4;         // dwords of one image (last holds 3 bytes: ROWB % 4 == 3 for odd V)
  static constexpr int ENTW = (NW + 1 + 3) / 4 * 4; // table entry: NW stream dwords + 1 visibility dword, 16-B multiple
  static_assert((ROWB & 3) == 3, "odd view sizes only");
};

struct Params {
  int32_t W, H, max_steps, see_through, competitive, shaping, auto_reset, dd_bytes;
  int32_t op, random_actions, PW, PH, n_entries;
  int32_t nt_obs;              // bit a: agent a's obs images leave with non-temporal stores (launch()'s streaming rule)
  int64_t n_envs, stride, n_chunks;
  uint64_t gid_base, t;
  uint32_t seed_lo, seed_hi;
  const uint32_t* map_pad;     // [PH][PW]
  const double* dist_lut;      // [H][W] distance to the shaper's goal cell
  const uint32_t* view_table;  // [n_entries][ENTW] (table path) : static view of (free cell, dir)
  const uint16_t* cell_index;  // [H][W] -> free-cell index (table path)
  const uint32_t* actions;     // u8x4 per env
  const uint32_t* order;       // u8x4 per env
  const uint32_t* respawn;     // int8x4 per (env, agent)
  const uint32_t* reset_spawn; // int8x4 per (env, agent)
  const uint8_t* mask;
  uint32_t* actions_out;       // u8x4 per env
  uint8_t* obs_image;
  double* reward;
  double* env_reward;
  uint8_t* dir_done;           // [stride][dd_bytes]: dir of agent a at byte a, done at byte A
  uint32_t* env_state;         // [stride][A+1]
  uint32_t* err;
  // mutable-grid kernel only (mg_dyn.h)
  int32_t kind, GW;            // MG_KIND_*, grid dwords per env
  int32_t inv_w20, GB;         // ceil(2^20 / W): cell index -> row by multiply and shift; GB = ceil(GW / 4) 16-byte row groups (mg_dyn.h)
  const uint32_t* base_grid;   // [GW] kind ids of the layout every reset starts from
  const uint32_t* grid0;       // mg_set_maps: [n_chunks][GW][64] per-env layouts every reset starts from (else null)
  const uint32_t* objdrop;     // int8x4 per (env, agent)
  const uint32_t* reset_gen;   // int8x4 per env
  uint32_t* carry;             // [stride]
  uint32_t* grid;              // [n_chunks][GW][64]
  int32_t n_goals, n_clutter, gen_flags, full_obs;   // scatter generators (CLUTTERED, MEETUP)
  int32_t gx, gy, LW, LH;      // shaper goal and the dims of the mutable-grid kernel's |dx|,|dy| distance LUT
  int32_t fixed_mode, fixed_pad;   // MG_FIXED_*: agents (re)start on fixed cells (shared-map kernels)
  uint64_t fixed_pd;           // their start cells, packed like pd (x | y<<6 per agent, 14 bits each)
  uint32_t fixed_pd_hi, roll_split;    // the fifth agent's (five agents do not fit one u64); roll_split: mg_dyn_rollout_kernel deals a chunk's agents out over waves
  const uint16_t* scatter;     // int8x2 per (env, scattered object): injected layout draws
  uint32_t* aux;               // [stride][4]  MEETUP goal positions + past distances
  uint8_t* obs_full;           // [stride][W*H*3]  fully observed obs
  // fused K-step rollout (mg_rollout_random): step k of the launch writes slot k of the caller's trajectory buffers
  int32_t n_roll, reward_f32;  // reward_f32: the reward buffers hold float, not double (mg_config.reward_dtype)
  uint8_t* traj_obs;           // [K][A][stride][V*V*3]
  double* traj_reward;         // [K][A][stride]
  uint8_t* traj_dd;            // [K][stride][dd_bytes]
  uint32_t* traj_actions;      // [K][stride][slot words], may be null
};

struct U4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al. SC'11).  Counter: (global env id, t lo, t hi,
// purpose<<28 | agent<<20 | attempt); key = seed.  Same definition as oracle/mg_oracle.c.
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a high and a low multiply, both quarter rate
    const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t h0 = (uint32_t)(m0 >> 32), l0 = (uint32_t)m0;
    const uint32_t h1 = (uint32_t)(m1 >> 32), l1 = (uint32_t)m1;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// MiniGridEnv._reward() = 1 - 0.9 * (step_count / max_steps) in IEEE
// float64 with one rounding per operation (-ffp-contract=off; hipcc's f64 division is the
// correctly rounded v_div_scale/v_div_fmas/v_div_fixup sequence).  Checked against CPython's
// value for every step count in tests/test_gpu_parity.py::test_device_reward_values.
__device__ __forceinline__ double device_reward(uint32_t step, int max_steps) {
  const double q = (double)step / (double)max_steps;
  const double m = 0.9 * q;
  return 1.0 - m;
}

// Per-agent values that are selected with a RUN-TIME agent id live in packed integers and are
// extracted with shifts: pd = (x:6 | y:6 | dir:2) of every agent, 14 bits each, in one u64;
// actions 3 bits each in one u32.  (A select chain over array elements gets turned by hipcc into
// an indexed scratch-memory load, whose s_waitcnt vmcnt would also drain the obs stores in flight.)
// Arrays below are only ever indexed with compile-time constants.
__device__ __forceinline__ uint32_t pd_get(uint64_t pd, int a) { return (uint32_t)(pd >> (14 * a)) & AW_POSDIR; }
__device__ __forceinline__ uint64_t pd_set(uint64_t pd, int a, uint32_t v) {
  return (pd & ~((uint64_t)AW_POSDIR << (14 * a))) | ((uint64_t)v << (14 * a));
}
// Five agents (Empty-16x16, FourRooms-v0, DoorKey-16x16) need 70 bits: the fifth agent's word rides in a second
// register.  pd_t<A> is the plain u64 for A <= 4, so those kernels compile exactly as before.
struct Pd5 { uint64_t lo; uint32_t hi; };
__device__ __forceinline__ uint32_t pd_get(const Pd5& pd, int a) { return a < 4 ? pd_get(pd.lo, a) : (pd.hi & AW_POSDIR); }
__device__ __forceinline__ Pd5 pd_set(Pd5 pd, int a, uint32_t v) {
  if (a < 4) pd.lo = pd_set(pd.lo, a, v); else pd.hi = v;
  return pd;
}
template <int A> struct PdOf { typedef uint64_t type; };
template <> struct PdOf<5> { typedef Pd5 type; };
template <int A> using pd_t = typename PdOf<A>::type;
constexpr int kMaxAgents = 5;

// place_obj's acceptance test (multigrid.py:603-613) on the shared static map.
// occ = bit mask of agents whose position currently counts (agent_pos not None).
template <int A>
__device__ __forceinline__ bool cell_free(const Params& p, const uint32_t* map, pd_t<A> pd,
                                          uint32_t occ, int x, int y) {
  if ((unsigned)x >= (unsigned)p.W || (unsigned)y >= (unsigned)p.H) return false;
  const int pad = (p.PW - p.W) >> 1;
  if (!(map[(y + pad) * p.PW + x + pad] & F_FREE)) return false;
  const uint32_t pos = (uint32_t)x | ((uint32_t)y << 6);
  bool ok = true;
#pragma unroll
  for (int b = 0; b < A; ++b) ok = ok && !(((occ >> b) & 1u) && (pd_get(pd, b) & AW_POS) == pos);
  return ok;
}

// place_one_agent + place_agent_at_pos (multigrid.py:646-675): returns x | y<<6 | dir<<12, with
// MG_ERR_* bits (if any) in bits 16+.  inj != nullptr: the draw the reference accepted is
// injected (int8x4 = x, y, dir, 0).  Rare path (goal touch, reset).
template <int A>
__device__ __forceinline__ uint32_t place_agent(const Params& p, const uint32_t* map, pd_t<A> pd, uint32_t occ,
                                                int a, int purpose, const uint32_t* inj, int64_t n, uint32_t gid) {
  int x = -1, y = -1, dir = 0;
  uint32_t err = 0;
  if (inj) {
    const uint32_t w = inj[n * A + a];
    x = (int8_t)(w & 255u); y = (int8_t)((w >> 8) & 255u); dir = (int8_t)((w >> 16) & 255u);
    if (x < 0) { err = MG_ERR_MISSING_DRAW; x = -1; }
    else if (!cell_free<A>(p, map, pd, occ, x, y) || (unsigned)dir > 3u) { err = MG_ERR_BAD_RESPAWN; x = -1; }
  } else {
    for (int k = 0; k < kMaxAttempts; ++k) {   // while True: multigrid.py:589-619 (bounded here)
      const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32),
                          ((uint32_t)purpose << 28) | ((uint32_t)a << 20) | (uint32_t)k, p.seed_lo, p.seed_hi);
      const int cx = (int)__umulhi(r.x, (uint32_t)p.W), cy = (int)__umulhi(r.y, (uint32_t)p.H);
      if (cell_free<A>(p, map, pd, occ, cx, cy)) { x = cx; y = cy; dir = (int)(r.z >> 30); break; }
    }
  }
  if (x < 0) {   // exit condition of the sampling loop: first free cell in storage order, dir 0
    dir = 0;
    for (int j = 0; j < p.H && x < 0; ++j)
      for (int i = 0; i < p.W; ++i)
        if (cell_free<A>(p, map, pd, occ, i, j)) { x = i; y = j; break; }
    if (x < 0) { x = 0; y = 0; }
  }
  return (uint32_t)x | ((uint32_t)y << 6) | ((uint32_t)dir << 12) | (err << 16);
}

// ---------------------------------------------------------------------------------------------
// Egocentric view (gen_obs_grid + encode, multigrid.py:997-1035, 1081).
//
// Closed form of slice + rotate_left x (dir+1): view cell (vx,vy) is world cell
//   pos + f*(V-1-vy) + r*(vx - V/2),  f = DIR_TO_VEC[dir], r = (-f.y, f.x)
// (equivalent to get_view_coords, multigrid.py:723-741).  Agents are transparent, so the static
// part of a view (cell codes with visibility applied, own cell = empty) depends only on
// (pos, dir): `static_view` computes it from the map; on the table path it was computed once on
// the host for every (free cell, dir) and is simply fetched from LDS.
// Output: s[0..NW-1] = the V*V 3-byte codes as a dword stream, vm = visibility bits (vy*V+vx).
// ---------------------------------------------------------------------------------------------
// One more cell of a view row: the opacity bits of row vy are collected by walking vx from V-1 down to 0, each step
// shifting the row mask left and taking the cell's F_OPAQUE (bit 31) as the new bit 0 — a single v_alignbit_b32.
__device__ __forceinline__ uint32_t view_row_bit(uint32_t row, uint32_t cell_word) {
  return __builtin_amdgcn_alignbit(row, cell_word, 31);
}

// process_vis + own cell + encode: from the V*V cell words (w[vx*V+vy]: type | color<<8 | state<<16 in the low three
// bytes, the top byte is ignored) and the per-row opacity bits (opq[vy] bit vx) to the dword stream
// and the visibility mask vm (bit vy*V + vx).  Four 3-byte codes make three dwords, one v_perm_b32 each.
// DEC (mutable-grid kernels, mg_dyn.h): the cell words are raw products of cell codes instead — after the mask
// (state = 0, type, 0, colour) from byte 0 up — so the visibility mask also strips them and the selectors differ.
template <int V, bool DEC = false>
__device__ __forceinline__ void pack_view(uint32_t (&w)[V * V], const uint32_t (&opq)[V], bool see_through,
                                          uint32_t own_code, uint32_t (&s)[ViewGeom<V>::NW + 1], uint64_t& vm) {
  constexpr int NW = ViewGeom<V>::NW, NC = V * V;
  static_assert(NC % 4 == 1 && NW == (NC / 4) * 3 + 1, "groups of four cells, then the last cell alone");
  uint32_t vis[V];
  if (see_through) {
#pragma unroll
    for (int j = 0; j < V; ++j) vis[j] = (1u << V) - 1u;
  } else {
    mg_vis_rows<V>(opq, vis);
  }
  w[(V / 2) * V + (V - 1)] = own_code;
  vm = 0;
#pragma unroll
  for (int j = 0; j < V; ++j) vm |= (uint64_t)vis[j] << (j * V);
#pragma unroll
  for (int c = 0; c < NC; ++c) {                                 // unseen -> (0,0,0)
    const int vx = c / V, vy = c % V;
    w[c] &= (uint32_t)__builtin_amdgcn_sbfe((int)vis[vy], vx, 1) & (DEC ? 0x07000F00u : 0xFFFFFFFFu);
  }
  // (type, colour, state) of a cell word are its bytes (0, 1, 2), or (1, 3, 0) with DEC
  constexpr uint32_t kSel0 = DEC ? 0x05000301u : 0x04020100u;    // c0.t c0.c c0.s c1.t
  constexpr uint32_t kSel1 = DEC ? 0x07050003u : 0x05040201u;    // c1.c c1.s c2.t c2.c
  constexpr uint32_t kSel2 = DEC ? 0x04070500u : 0x06050402u;    // c2.s c3.t c3.c c3.s
#pragma unroll
  for (int q = 0; q < NC / 4; ++q) {
    const uint32_t c0 = w[4 * q], c1 = w[4 * q + 1], c2 = w[4 * q + 2], c3 = w[4 * q + 3];
    s[3 * q] = __builtin_amdgcn_perm(c1, c0, kSel0);
    s[3 * q + 1] = __builtin_amdgcn_perm(c2, c1, kSel1);
    s[3 * q + 2] = __builtin_amdgcn_perm(c3, c2, kSel2);
  }
  s[NW - 1] = DEC ? __builtin_amdgcn_perm(0u, w[NC - 1], 0x0C000301u) : (w[NC - 1] & 0xFFFFFFu);
  s[NW] = 0;
}

template <int V>
__device__ __forceinline__ void static_view(const Params& p, const uint32_t* map, uint32_t me,
                                            uint32_t (&s)[ViewGeom<V>::NW + 1], uint64_t& vm) {
  constexpr int NW = ViewGeom<V>::NW;
  const int pad = (p.PW - p.W) >> 1;
  const int x = aw_x(me), y = aw_y(me), d = aw_d(me);
  const int fx = (d == 0) - (d == 2), fy = (d == 1) - (d == 3);
  const int rx = -fy, ry = fx;
  const int dr = rx + p.PW * ry, df = fx + p.PW * fy;
  const int i0 = (y + pad + fy * (V - 1) - ry * (V / 2)) * p.PW + (x + pad + fx * (V - 1) - rx * (V / 2));
  uint32_t code[V * V];
  uint32_t opq[V];
#pragma unroll
  for (int j = 0; j < V; ++j) opq[j] = 0;
#pragma unroll
  for (int vx = V - 1; vx >= 0; --vx)
#pragma unroll
    for (int vy = 0; vy < V; ++vy) {
      const uint32_t w = map[i0 + vx * dr - vy * df];
      code[vx * V + vy] = w;
      opq[vy] = view_row_bit(opq[vy], w);
    }
  // own cell <- carried object or None, after process_vis (multigrid.py:1029-1033); nothing can
  // be carried on a static map
  pack_view<V>(code, opq, p.see_through != 0, (uint32_t)MG_T_EMPTY, s, vm);
}

template <int V>
__device__ __forceinline__ void table_view(const Params& p, const uint32_t* table, const uint16_t* cidx, uint32_t me,
                                           uint32_t (&s)[ViewGeom<V>::NW + 1], uint64_t& vm) {
  constexpr int NW = ViewGeom<V>::NW, ENTW = ViewGeom<V>::ENTW;
  const int e = (int)cidx[aw_y(me) * p.W + aw_x(me)] * 4 + aw_d(me);
  const uint4* row = reinterpret_cast<const uint4*>(table + e * ENTW);
  uint32_t t[ENTW];
#pragma unroll
  for (int i = 0; i < ENTW / 4; ++i) {
    const uint4 q = row[i];
    t[4 * i] = q.x; t[4 * i + 1] = q.y; t[4 * i + 2] = q.z; t[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) s[i] = t[i];
  s[NW] = 0;
  vm = t[NW];
}

// One 16-byte obs vector to HBM.  nt (wave-uniform, from the launch) = non-temporal: the line is not kept in L2 / the
// Infinity Cache.  A batch whose outputs exceed the cache is rewritten every step without ever being re-read in
// between, so keeping the lines only costs evictions (4M envs: 216 -> 192 us); a launch-resident batch is faster
// with ordinary stores (1M envs, table kernel: 51.5 vs 57.7 us), hence a run-time choice (launch()).
__device__ __forceinline__ void store_vec16(uint4* dst, const uint4 v, bool nt) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  if (nt) __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4*>(dst));
  else *dst = v;
}

// Stage the 64 images of this wave in LDS and stream them to HBM as one contiguous chunk.
// DOOR (mutable-grid kernels that decode cell codes arithmetically, mg_dyn.h): doorw = x | y<<6 | state<<12 | colour<<16
// of the one closed or locked door the view cannot decode from its cell code (0xFFFFFFFF: none), patched in like an agent.
template <int A, int V, bool DOOR = false>
__device__ __forceinline__ void emit_obs(pd_t<A> pd, int a, uint32_t (&s)[ViewGeom<V>::NW + 1], uint64_t vm,
                                         uint32_t* stage, int lane, uint8_t* gchunk, bool nt, uint8_t* gchunk2 = nullptr,
                                         uint32_t doorw = 0xFFFFFFFFu) {
  constexpr int ROWB = ViewGeom<V>::ROWB, NW = ViewGeom<V>::NW;
  const uint32_t me = pd_get(pd, a);
  const int x = aw_x(me), y = aw_y(me), d = aw_d(me);
  const int fx = (d == 0) - (d == 2), fy = (d == 1) - (d == 3);
  const int rx = -fy, ry = fx;
  // This lane's image occupies bytes [lane*ROWB, lane*ROWB+ROWB) of the wave chunk, which is not
  // dword aligned.  A lane owns the chunk dwords that START inside its byte range; the last of
  // them is completed with the first bytes of the next lane's stream.  (gfx950's LDS also takes unaligned 16-byte
  // stores, which would save the shuffle and the byte alignment: measured in round 3, the static kernel went from
  // 50.5 to 57.5 us and DoorKey-8x8 from 137.5 to 163 us with them.)
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
  uint8_t* sb = reinterpret_cast<uint8_t*>(stage) + b0;
  // other agents: patch their cell (3 bytes) if it lies in the view and is visible; encoded
  // (10, id, (dir_other - dir_obs - 1) mod 4)  (Agent.encode :137-139, rotate_left :289-295)
#pragma unroll
  for (int b = 0; b < A; ++b) {
    if (b == a) continue;
    const uint32_t o = pd_get(pd, b);
    const int ox = aw_x(o) - x, oy = aw_y(o) - y;
    const int along = ox * fx + oy * fy, side = ox * rx + oy * ry;
    const int vy = V - 1 - along, vx = side + V / 2;
    if ((unsigned)vx < (unsigned)V && (unsigned)vy < (unsigned)V) {
      if ((vm >> ((uint32_t)__mul24(vy, V) + (uint32_t)vx & 63u)) & 1u) {      // (32-bit index arithmetic, not widened)
        uint8_t* q = sb + ((uint32_t)__mul24(__mul24(vx, V) + vy, 3) & 255u);
        q[0] = (uint8_t)MG_T_AGENT;
        q[1] = (uint8_t)b;
        q[2] = (uint8_t)(((uint32_t)aw_d(o) - (uint32_t)d - 1u) & 3u);
      }
    }
  }
  if (DOOR && doorw != 0xFFFFFFFFu) {        // Door.encode of a door that is not open: (4, colour, 1 closed | 2 locked)
    const int ox = aw_x(doorw) - x, oy = aw_y(doorw) - y;
    const int along = ox * fx + oy * fy, side = ox * rx + oy * ry;
    const int vy = V - 1 - along, vx = side + V / 2;
    if ((unsigned)vx < (unsigned)V && (unsigned)vy < (unsigned)V) {
      if ((vm >> ((uint32_t)__mul24(vy, V) + (uint32_t)vx & 63u)) & 1u) {      // (32-bit index arithmetic, not widened)
        uint8_t* q = sb + ((uint32_t)__mul24(__mul24(vx, V) + vy, 3) & 255u);
        q[0] = (uint8_t)MG_T_DOOR;
        q[1] = (uint8_t)((doorw >> 16) & 7u);
        q[2] = (uint8_t)((doorw >> 12) & 3u);
      }
    }
  }
  // LDS is processed in order per wave, so the reads below see the writes above; the compiler
  // only needs to be kept from reordering them (cross-lane dependence it cannot see).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int NV = 4 * ROWB;                // 16-byte vectors in the 64-image chunk
  constexpr int NI = (NV + kWave - 1) / kWave; // ... per lane; only the last may fall past the chunk
  const uint4* src = reinterpret_cast<const uint4*>(stage);
  uint4* dst = reinterpret_cast<uint4*>(gchunk);
  // all the LDS reads first, then the stores, and the (wave-uniform) choice of store flavour outside the loops: one LDS
  // round trip per image chunk instead of one per vector
  uint4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = i * kWave + lane;
    v[i] = src[(i + 1) * kWave <= NV || idx < NV ? idx : NV - 1];      // (a clamped read, not a masked one: one ds_read_b128)
  }
#if MG_ABL == 1
#pragma unroll
  for (int i = 0; i < NI; ++i) asm volatile("" :: "v"(v[i].x), "v"(v[i].y), "v"(v[i].z), "v"(v[i].w));
#else
  if (nt) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * kWave + lane;
      if ((i + 1) * kWave <= NV || idx < NV) store_vec16(dst + idx, v[i], true);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * kWave + lane;
      if ((i + 1) * kWave <= NV || idx < NV) store_vec16(dst + idx, v[i], false);
    }
  }
#endif
  if (gchunk2) {                               // wave-uniform; only the fused rollout ever passes one
    uint4* dst2 = reinterpret_cast<uint4*>(gchunk2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * kWave + lane;
      if ((i + 1) * kWave <= NV || idx < NV) store_vec16(dst2 + idx, v[i], false);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inputs of one chunk, fetched one iteration ahead of their use (software prefetch: vmcnt
// retires in issue order, so loads issued BEFORE the previous chunk's stores do not wait for them).
// Inputs of one chunk, requested one iteration ahead of their use.  The loads are inline asm so
// that hipcc neither waits for them at once nor drains the obs stores in flight when the values
// are finally used: vmcnt retires vector-memory operations in issue order, so a counted
// s_waitcnt vmcnt(K) issued after >= K younger stores retires these loads without waiting for
// those stores (wait_chunk).  Nothing is computed from the registers before that wait.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// action / order slots per env: four bytes (one dword) up to four agents, eight bytes for five
template <int A> constexpr int kSlotWords = A > 4 ? 2 : 1;
struct ChunkIn {
  u32x4 q;                       // A == 3: the four env_state words
  uint32_t w0, w1, w2, w3, w4, w5;   // other A: env_state words (scalars, not an array: an array member made hipcc keep the
                                 // launch-resident batch's ChunkIn[] in scratch and store the registers there while
                                 // their loads were still in flight — tools/isa_prefetch_lint.py)
  uint32_t act;                  // the env's four action bytes, as loaded
  uint32_t act2;                 // five agents: action bytes 4..7 (the action / order slots are 8 bytes per env then)
};

__device__ __forceinline__ uint32_t asm_load_u32(const void* ptr) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(ptr) : "memory");
  return v;
}

// FAST = the hot configuration, known at compile time: a free-running step() whose actions come
// from HBM (no replay injection, no on-device action draw, no env_reward output).  Everything
// else (reset, replay, step_random, emit_env_reward) runs the general instantiation.
__device__ __forceinline__ void zero_chunk(ChunkIn& in) {
  in.q = u32x4{0, 0, 0, 0};
  in.w0 = in.w1 = in.w2 = in.w3 = in.w4 = in.w5 = 0u;
  in.act = in.act2 = 0x06060606u;
}

template <int A, bool FAST>
__device__ __forceinline__ void load_chunk(const Params& p, int64_t env0, int lane, ChunkIn& in) {
  const int64_t n = env0 + lane;
  zero_chunk(in);
  if (A == 3) {
    const void* ptr = reinterpret_cast<const uint4*>(p.env_state) + n;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(in.q) : "v"(ptr) : "memory");
  } else {
    const uint32_t* sp = p.env_state + n * (A + 1);
    in.w0 = asm_load_u32(sp);
    if (A >= 1) in.w1 = asm_load_u32(sp + 1);
    if (A >= 2) in.w2 = asm_load_u32(sp + 2);
    if (A >= 3) in.w3 = asm_load_u32(sp + 3);
    if (A >= 4) in.w4 = asm_load_u32(sp + 4);
    if (A >= 5) in.w5 = asm_load_u32(sp + 5);
  }
  if (FAST || (p.op == OP_STEP && !p.random_actions)) {
    in.act = asm_load_u32(p.actions + n * kSlotWords<A>);
    if (A > 4) in.act2 = asm_load_u32(p.actions + n * kSlotWords<A> + 1);
  }
}

// Retire the loads of load_chunk: at most K vector-memory operations younger than them may stay
// in flight.  The registers are operands so that no use (not even a copy) can be moved above it.
template <int K>
__device__ __forceinline__ void wait_chunk(ChunkIn& in) {
  asm volatile("s_waitcnt vmcnt(%9)"
               : "+v"(in.q), "+v"(in.w0), "+v"(in.w1), "+v"(in.w2), "+v"(in.w3), "+v"(in.w4), "+v"(in.w5),
                 "+v"(in.act), "+v"(in.act2)
               : "n"(K)
               : "memory");
}

// ONE Philox block per free-running env-step (round 3; it used to be two).  Words x, y, z of the block keyed P_ORDER feed
// the Fisher-Yates of the agent order (A - 1 draws); word w feeds the "random actions" of the synthetic rollouts: agent a's
// action is the a-th base-7 digit of w / 2^32 — multiply-high by 7, carry on with the low half of the product.  7 is odd, so
// the low half is as uniform as w, and A <= 4 digits (7^4 << 2^32) are as exact as one multiply-high each.  Five agents need
// all four words for the order; their actions are the five digits of word x of a second block (P_ACTION).  The order of a step
// that draws no actions (mg_step, mg_step_replay) is unchanged by this.
__device__ __forceinline__ U4 step_block(const Params& p, uint32_t gid, int purpose) {
  return philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32), (uint32_t)purpose << 28, p.seed_lo, p.seed_hi);
}

// uniform in [0,6] per agent, 3 bits each; also stored to actions_out in the layout mg_step reads.
template <int A>
__device__ __forceinline__ uint32_t draw_actions(const Params& p, int64_t n, uint32_t gid, bool active, const U4& blk) {
  uint32_t w = A > 4 ? step_block(p, gid, P_ACTION).x : blk.w;
  uint32_t actw = 0;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    actw |= __umulhi(w, 7u) << (3 * a);
    w *= 7u;
  }
  if (p.actions_out && active) {
    uint32_t o = 0;
#pragma unroll
    for (int a = 0; a < A && a < 4; ++a) o |= ((actw >> (3 * a)) & 7u) << (8 * a);
    p.actions_out[n * kSlotWords<A>] = o;
    if (A > 4) p.actions_out[n * kSlotWords<A> + 1] = (actw >> 12) & 7u;
  }
  return actw;
}

// Agent order of this step (multigrid.py:973-974), packed kOrdBits<A> bits per slot (2; 3 for five agents):
// injected, or a Fisher-Yates on the step's Philox block.
template <int A> constexpr int kOrdBits = A > 4 ? 3 : 2;
template <int A, bool FAST>
__device__ __forceinline__ uint32_t agent_order(const Params& p, int64_t n, bool active, uint32_t& err, const U4& blk) {
  constexpr int OB = kOrdBits<A>;
  constexpr uint32_t OM = (1u << OB) - 1u;
  uint32_t ordw = 0;
#pragma unroll
  for (int k = 0; k < A; ++k) ordw |= (uint32_t)k << (OB * k);
  if (!FAST && p.order) {
    uint32_t seen = 0, w = 0;
    const uint32_t ow = p.order[n * kSlotWords<A>], ow2 = A > 4 ? p.order[n * kSlotWords<A> + 1] : 0u;
#pragma unroll
    for (int k = 0; k < A; ++k) {
      const uint32_t o = ((k < 4 ? ow >> (8 * k) : ow2 >> (8 * (k - 4)))) & 255u;
      if (o < (uint32_t)A) seen |= 1u << o;
      w |= (o & OM) << (OB * k);
    }
    if (seen != (1u << A) - 1u) { if (active) err |= MG_ERR_BAD_ORDER; }
    else ordw = w;
  } else if (A > 1 && MG_ABL != 3) {
    const uint32_t rr[4] = {blk.x, blk.y, blk.z, blk.w};
    // Fisher-Yates from the top, as np.random.shuffle does: swap slots i and j = rand(i+1)
#pragma unroll
    for (int i = A - 1, w = 0; i >= 1; --i, ++w) {
      const uint32_t j = __umulhi(rr[w & 3], (uint32_t)(i + 1));
      const uint32_t oi = (ordw >> (OB * i)) & OM, oj = (ordw >> (OB * j)) & OM;
      ordw = (ordw & ~(OM << (OB * i)) & ~(OM << (OB * j))) | (oi << (OB * j));
      ordw = (ordw & ~(OM << (OB * i))) | (oj << (OB * i));
    }
  }
  return ordw;
}

// shape_rewards for one agent, float64, one rounding per operation: the v8 shaper (v8_robust_ppo.py:171-220) or, with
// shaping == MG_SHAPING_V7, the v7 one (v7_stable_goal_focused_ppo.py:60-108; same structure, other constants, and a
// goal touch leaves prev_* and the count as they were, v7:70-72, where v8 clears them, v8:181-186).
// w = the agent's shaper bits (prev_x, prev_y, valid, stationary), me = its post-step position word, env_rew = its
// env reward.  Returns the new shaper bits.
// REL = false: dist[y*W + x] = distance of cell (x, y) to the handle's one goal (shared-map kernels);
// REL = true: dist[|dy|*LW + |dx|] = sqrt(dx^2 + dy^2), looked up relative to goal (gx, gy), which may
// differ per env (mutable-grid kernel).  Both tables are built on the host with the same sqrt.
template <bool REL = false>
__device__ __forceinline__ uint32_t v8_shape(const Params& p, const double* dist, uint32_t w, uint32_t me, uint32_t act,
                                             double env_rew, double& out, int gx = 0, int gy = 0) {
  const bool v7 = p.shaping == MG_SHAPING_V7;
  if (env_rew > 0.0) {                            // goal touch: prev_* cleared, count 0 (v8:181-186); kept (v7:70-72)
    out = v7 ? 10.0 : 5.0;
    return v7 ? (w & ~AW_POSDIR) : 0u;
  }
  const int x = aw_x(me), y = aw_y(me);
  double r = 0.0;
  const bool has = (w >> 26) & 1u;
  const int px = (int)((w >> 14) & 63u), py = (int)((w >> 20) & 63u);
  uint32_t stat = (w >> 27) & 7u;
  const double cur = REL ? dist[abs(y - gy) * p.LW + abs(x - gx)] : dist[y * p.W + x];   // np.linalg.norm(pos - goal), v8:192
  if (has) {
    const int qx = min(px, p.W - 1), qy = min(py, p.H - 1);
    const double prev = REL ? dist[abs(qy - gy) * p.LW + abs(qx - gx)] : dist[qy * p.W + qx];
    r = r + (prev - cur) * 0.2;                   // v8:193-195, v7:79-81
    if (px == x && py == y) {                     // v8:199-206, v7:85-93
      stat = min(stat + 1u, 7u);                  // v8 tests "> 3" only; v7 also takes min(0.05, count * 0.01), which is
      if (stat > 3u) {                            // 0.05 from count 5 on: saturating at 7 is exact for both
        const double pen = (double)stat * 0.01;
        r = r - ((v7 && pen < 0.05) ? pen : 0.05);
      }
    } else {
      stat = 0u;
      r = r + (v7 ? 0.01 : 0.02);
    }
  }
  if (act == 2u) r = r + 0.02;                    // v8:209-212, v7:96-99
  else if (act <= 1u) r = r + (v7 ? 0.005 : 0.01);
  if (-0.2 > r) r = -0.2;                         // max(reward, -0.2), v8:215, v7:104
  out = r;
  return ((uint32_t)x << 14) | ((uint32_t)y << 20) | (1u << 26) | (stat << 27);   // v8:218, v7:82,106
}

// EmptyEnv.place_one_agent with agent_start='fixed' (multigrid_envs/empty.py:72-104): agent a goes to its start cell
// facing east; whoever stands there is sent back to its own start cell first, and so on down the chain.  on_grid =
// agents that currently have a cell (the finishing agent was taken off the grid before, multigrid.py:817-819).
__device__ __forceinline__ uint32_t fixed_start(const Params& p, int a) {   // start cell of agent a (x | y<<6)
  return (a < 4 ? pd_get(p.fixed_pd, a) : p.fixed_pd_hi) & AW_POS;
}

template <int A>
__device__ __forceinline__ pd_t<A> fixed_place(const Params& p, pd_t<A> pd, int a, uint32_t on_grid) {
  int t = a;
#pragma unroll 1
  for (int it = 0; it < A; ++it) {
    const uint32_t spos = fixed_start(p, t);
    int b = -1;
#pragma unroll
    for (int c = 0; c < A; ++c)
      if (c != t && ((on_grid >> c) & 1u) && (pd_get(pd, c) & AW_POS) == spos) b = c;
    pd = pd_set(pd, t, spos);                       // agent_start_dir = 0
    on_grid |= 1u << t;
    if (b < 0) break;
    t = b;
  }
  return pd;
}

// One step()/reset() of one chunk (64 envs) whose state is in registers: pd = (x, y, dir) of every agent, shw = the shaper
// bits of the agent words, st = step_count | done bits << 16, actw = the actions (3 bits per agent).  The outputs go to
// p.obs_image / p.reward / p.dir_done (and p.env_state when store_state); ROLL = a step of the fused rollout
// (mg_rollout_kernel): free-running like FAST, actions drawn on the device, and on its last step a second set of
// destinations (the handle's own buffers, *2 below) besides the trajectory slot.
// SPLIT (mg_rollout_split_kernel, small batches): two to A waves share a chunk; every one of them runs the whole action phase —
// it is serial in the drawn order and cheap — and then does its own agents' share of the rest (`mine`, a wave-uniform bit per
// agent): their shaper, reward store, view and image, their words of env_state at the end; the wave of agent 0 also stores
// dir_done (and the actions, by way of p.actions_out being null for the others) and the step word.  shw[] of the other agents
// is not maintained by this wave.
template <int A, int V, bool TAB, bool FAST, bool ROLL = false, bool SPLIT = false>
__device__ __forceinline__ void step_chunk(const Params& p, const uint32_t* map, const double* dist, const uint32_t* table,
                                           const uint16_t* cidx, uint32_t* stage, int lane, int64_t env0,
                                           pd_t<A>& pd, uint32_t (&shw)[A], uint32_t& st, uint32_t actw, bool store_state,
                                           uint32_t& err_acc, uint8_t* obs2 = nullptr, double* reward2 = nullptr,
                                           uint8_t* dd2 = nullptr, uint32_t mine = ~0u) {
  static_assert(!ROLL || FAST, "the rollout is a free-running step");
  static_assert(!SPLIT || ROLL, "agents are dealt out over waves in the fused rollout only");
  constexpr int ROWB = ViewGeom<V>::ROWB, NW = ViewGeom<V>::NW;
  const int64_t n = env0 + lane;
  const bool active = n < p.n_envs;
  const uint32_t gid = (uint32_t)(p.gid_base + (uint64_t)n);
  const int pad = (p.PW - p.W) >> 1;
  uint32_t err = 0;
  uint32_t done_flag = 0;

  const bool is_step = FAST || p.op == OP_STEP;
  const uint32_t* const inj_respawn = FAST ? nullptr : p.respawn;
  const uint32_t* const inj_spawn = FAST ? nullptr : p.reset_spawn;
  if (is_step) {
    // ---- actions -------------------------------------------------------------------------
    if (ROLL && p.actions) {                   // mg_rollout with the caller's action ring: step k's slot, decoded as mg_step does
      const uint32_t w = p.actions[n * kSlotWords<A>], w2 = A > 4 ? p.actions[n * kSlotWords<A> + 1] : 0u;
      actw = 0;
#pragma unroll
      for (int a = 0; a < A; ++a) actw |= min(((a < 4 ? w >> (8 * a) : w2 >> (8 * (a - 4)))) & 255u, 7u) << (3 * a);
    }
    // ---- the step's one Philox block: agent order, and the actions of a synthetic rollout ------
    const bool draw_act = ROLL ? (p.actions == nullptr) : (!FAST && p.random_actions != 0);      // wave-uniform
    U4 blk = U4{0u, 0u, 0u, 0u};
    if (MG_ABL != 3 && (draw_act || (A > 1 && (FAST || !p.order)))) blk = step_block(p, gid, P_ORDER);
    if (draw_act) actw = draw_actions<A>(p, n, gid, active, blk);
    // ---- step_count += 1 (multigrid.py:968) ----------------------------------------------
    uint32_t step = (st & 0xFFFFu) + 1u;
    uint32_t dbits = st >> 16;
    if (step > 0xFFFFu) step = 0xFFFFu;

    // ---- agent order (multigrid.py:973-974), packed 2 bits per slot --------------------------
    const uint32_t ordw = agent_order<A, FAST>(p, n, active, err, blk);

    // ---- step_one_agent in that order (multigrid.py:860-924), branch-free except respawn -----
    uint32_t gbits = 0;          // agents whose forward touched the goal this step
#pragma unroll 1
    for (int k = 0; k < A; ++k) {
      const int a = (int)((ordw >> (kOrdBits<A> * k)) & ((1u << kOrdBits<A>) - 1u));
      const uint32_t s = pd_get(pd, a);
      const uint32_t act = (actw >> (3 * a)) & 7u;
      const int x = aw_x(s), y = aw_y(s), d = aw_d(s);
      const int fx = x + (d == 0) - (d == 2), fy = y + (d == 1) - (d == 3);   // front_pos, pre-action
      const uint32_t fpos = (uint32_t)fx | ((uint32_t)fy << 6);
      bool blocked = false;                                                   // _forward :910-913
#pragma unroll
      for (int b = 0; b < A; ++b) blocked = blocked || (b != a && (pd_get(pd, b) & AW_POS) == fpos);
      const uint32_t cell = map[(fy + pad) * p.PW + fx + pad];
      const bool go = (act == 2u) && !blocked;
      const int nd = (act == 0u) ? ((d + 3) & 3) : (act == 1u) ? ((d + 1) & 3) : d;   // :867-876
      uint32_t npd = (s & AW_POS) | ((uint32_t)nd << 12);
      if (go && (cell & F_FREE)) npd = fpos | ((uint32_t)d << 12);            // move_agent :834-847
      // (a wall ahead: nothing moves, still "successful"; pickup/drop/toggle/done: no-ops here)
      if (go && (cell & F_KILL)) {                                            // goal or lava: agent_is_done :815-832
        dbits |= 1u << a;
        if (cell & F_GOAL) gbits |= 1u << a;                                  // cell re-read is still the goal :881-883
        if (p.fixed_mode == MG_FIXED_ALWAYS) {
          pd = fixed_place<A>(p, pd, a, ((1u << A) - 1u) & ~(1u << a));
          npd = pd_get(pd, a);
        } else {
          const uint32_t w = place_agent<A>(p, map, pd, ((1u << A) - 1u) & ~(1u << a), a, P_RESPAWN, inj_respawn, n, gid);
          npd = w & AW_POSDIR;
          err |= w >> 16;
        }
      }
      if (act > 6u) err |= MG_ERR_BAD_ACTION;                                 // assert False :902
      pd = pd_set(pd, a, npd);
    }
    // _reward() (multigrid.py:883): the same value for every agent that touched the goal this step
    double rgoal = 0.0;
    if (gbits) rgoal = device_reward(step, p.max_steps);

    // ---- collective done (multigrid.py:986-993) -------------------------------------------
    done_flag = (p.competitive && dbits != 0u) || (step >= (uint32_t)p.max_steps);

    // ---- v8 shape_rewards (v8_robust_ppo.py:171-220), float64, one rounding per operation ---
    double rew[A], out_rew[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      rew[a] = ((gbits >> a) & 1u) ? rgoal : 0.0;
      out_rew[a] = rew[a];
      if (SPLIT && !((mine >> a) & 1u)) continue;             // (wave-uniform)
      if (p.shaping != MG_SHAPING_NONE && MG_ABL != 4)
        shw[a] = v8_shape(p, dist, shw[a], pd_get(pd, a), (actw >> (3 * a)) & 7u, rew[a], out_rew[a]);
    }

    st = step | (dbits << 16);
    if (active) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        if (SPLIT && !((mine >> a) & 1u)) continue;
        const int64_t i = (int64_t)a * p.stride + n;
        // (not in the FAST step instantiation: the wave-uniform branch alone cost the headline kernel 1.1 us of 50.8;
        //  a float32 handle steps through the general instantiation)
        if ((!FAST || ROLL) && p.reward_f32) {                // torch.FloatTensor(rewards) (v8:311), done at the source
          const float r32 = (float)out_rew[a];
          if (ROLL) __builtin_nontemporal_store(r32, reinterpret_cast<float*>(p.reward) + i);
          else reinterpret_cast<float*>(p.reward)[i] = r32;
          if (ROLL && reward2) reinterpret_cast<float*>(reward2)[i] = r32;
        } else {
          if (ROLL) __builtin_nontemporal_store(out_rew[a], p.reward + i); else p.reward[i] = out_rew[a];
          if (ROLL && reward2) reward2[i] = out_rew[a];
        }
        if (!FAST && p.env_reward) p.env_reward[i] = rew[a];
      }
    }
  }

  // ---- reset() (multigrid.py:466-501; cluttered.py:55-63): the explicit call, or the auto-reset
  //      of an env whose step just returned done (same launch) --------------------------------
  const bool do_reset = (!is_step) ? (p.mask ? (p.mask[active ? n : 0] != 0) : true)
                                   : (done_flag && p.auto_reset);
  if (do_reset) {
#pragma unroll 1
    for (int a = 0; a < A; ++a) {
      if (p.fixed_mode == MG_FIXED_ALWAYS) {
        pd = fixed_place<A>(p, pd, a, (1u << a) - 1u);
      } else if (p.fixed_mode == MG_FIXED_AT_RESET) {       // MazeEnv: place_agent_at_pos(start_pos), direction drawn (maze.py:86)
        const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32), ((uint32_t)P_SPAWN << 28) | ((uint32_t)a << 20),
                            p.seed_lo, p.seed_hi);
        pd = pd_set(pd, a, fixed_start(p, a) | ((r.z >> 30) << 12));
      } else {
        const uint32_t w = place_agent<A>(p, map, pd, (1u << a) - 1u, a, P_SPAWN, inj_spawn, n, gid);
        pd = pd_set(pd, a, w & AW_POSDIR);
        err |= w >> 16;
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) shw[a] = 0u;          // shaper state cleared (v8:227-229)
    st = 0u;
  }

  if (active) {
    if (SPLIT) {
      if (store_state) {
#pragma unroll
        for (int a = 0; a < A; ++a)
          if ((mine >> a) & 1u) p.env_state[n * (A + 1) + a] = shw[a] | pd_get(pd, a);
        if (mine & 1u) p.env_state[n * (A + 1) + A] = st;
      }
    } else if (store_state) {
      uint32_t wout[A];
#pragma unroll
      for (int a = 0; a < A; ++a) wout[a] = shw[a] | pd_get(pd, a);
      if (A == 3) {
        reinterpret_cast<uint4*>(p.env_state)[n] = make_uint4(wout[0], wout[A > 1 ? 1 : 0], wout[A > 2 ? 2 : 0], st);
      } else {
#pragma unroll
        for (int a = 0; a < A; ++a) p.env_state[n * (A + 1) + a] = wout[a];
        p.env_state[n * (A + 1) + A] = st;
      }
    }
    // obs['direction'][a] (multigrid.py:1083) for every agent + the collective done flag: one store
    if (!SPLIT || (mine & 1u)) {
      uint64_t dd = (uint64_t)done_flag << (8 * A);
#pragma unroll
      for (int a = 0; a < A; ++a) dd |= (uint64_t)aw_d(pd_get(pd, a)) << (8 * a);
      if (A + 1 <= 4) reinterpret_cast<uint32_t*>(p.dir_done)[n] = (uint32_t)dd;
      else reinterpret_cast<uint64_t*>(p.dir_done)[n] = dd;
      if (ROLL && dd2) {
        if (A + 1 <= 4) reinterpret_cast<uint32_t*>(dd2)[n] = (uint32_t)dd;
        else reinterpret_cast<uint64_t*>(dd2)[n] = dd;
      }
    }
    err_acc |= err;
  }

  // ---- gen_obs (multigrid.py:1037-1067) -----------------------------------------------------
#if MG_ABL != 2
#pragma unroll
  for (int a = 0; a < (MG_ABL == 8 ? 1 : A); ++a) {   // (8: timing-only, one agent's view per wave — what splitting the agents over waves could buy)
    if (SPLIT && !((mine >> a) & 1u)) continue;
    uint32_t s[NW + 1];
    uint64_t vm;
    if (TAB) table_view<V>(p, table, cidx, pd_get(pd, a), s, vm);
    else static_view<V>(p, map, pd_get(pd, a), s, vm);
    emit_obs<A, V>(pd, a, s, vm, stage, lane, p.obs_image + ((int64_t)a * p.stride + env0) * ROWB, ((p.nt_obs >> a) & 1) != 0,
                   (ROLL && obs2) ? obs2 + ((int64_t)a * p.stride + env0) * ROWB : nullptr);
  }
#endif
}

// The agent words / step word / actions of one chunk as loaded -> registers.
template <int A>
__device__ __forceinline__ void unpack_chunk(const Params& p, const ChunkIn& in, pd_t<A>& pd, uint32_t (&shw)[A], uint32_t& st,
                                             uint32_t& actw) {
  uint32_t win[6];
  if (A == 3) { win[0] = in.q.x; win[1] = in.q.y; win[2] = in.q.z; win[3] = in.q.w; win[4] = 0u; win[5] = 0u; }
  else {
    win[0] = in.w0; win[1] = in.w1; win[2] = in.w2; win[3] = in.w3; win[4] = in.w4; win[5] = in.w5;
  }
  pd = pd_t<A>{};               // x | y<<6 | dir<<12 of every agent, 14 bits each
  actw = 0;
  st = win[A];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    actw |= min(((a < 4 ? in.act >> (8 * a) : in.act2 >> (8 * (a - 4)))) & 255u, 7u) << (3 * a);
    // keep every later LDS index in range whatever the buffer holds
    const uint32_t w = win[a];
    const uint32_t cx = min(w & 63u, (uint32_t)(p.W - 1)), cy = min((w >> 6) & 63u, (uint32_t)(p.H - 1));
    pd = pd_set(pd, a, cx | (cy << 6) | (w & 0x3000u));
    shw[a] = w & ~AW_POSDIR;              // v8 shaper bits of the agent word (bits 14..29)
  }
}

// One chunk (64 envs) of one step()/reset().
template <int A, int V, bool TAB, bool FAST>
__device__ __forceinline__ void do_chunk(const Params& p, const uint32_t* map, const double* dist, const uint32_t* table,
                                         const uint16_t* cidx, uint32_t* stage, int lane, int64_t env0,
                                         const ChunkIn& in, uint32_t& err_acc) {
  pd_t<A> pd;
  uint32_t shw[A], st, actw;
  unpack_chunk<A>(p, in, pd, shw, st, actw);
  step_chunk<A, V, TAB, FAST>(p, map, dist, table, cidx, stage, lane, env0, pd, shw, st, actw, true, err_acc);
}

template <int A, int V, bool TAB, bool FAST>
__global__ __launch_bounds__(kMaxBlock) void mg_kernel(const Params p) {
  constexpr int ROWB = ViewGeom<V>::ROWB, ENTW = ViewGeom<V>::ENTW;
  extern __shared__ __align__(16) uint8_t smem[];
  // LDS carve: [stage: nw x 64*ROWB B][view table (TAB)][dist LUT: W*H f64][map: PW*PH u32][cell index (TAB)]
  const int nw = blockDim.x >> 6;
  uint32_t* stage_all = reinterpret_cast<uint32_t*>(smem);
  uint32_t* table = reinterpret_cast<uint32_t*>(smem + (size_t)nw * kWave * ROWB);
  double* dist = reinterpret_cast<double*>(table + (TAB ? p.n_entries * ENTW : 0));
  uint32_t* map = reinterpret_cast<uint32_t*>(dist + p.W * p.H);
  uint16_t* cidx = reinterpret_cast<uint16_t*>(map + p.PW * p.PH);

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  // persistent: each wave strides over the 64-env chunks on its own; the first chunk's inputs
  // are requested before the LDS tables are filled
  const int64_t cstride = (int64_t)gridDim.x * nw;
  int64_t c = (int64_t)blockIdx.x * nw + wv;
  ChunkIn cur, nxt;
  zero_chunk(cur);
  if (c < p.n_chunks) load_chunk<A, FAST>(p, c * kWave, lane, cur);

  if (TAB) {
    const uint4* src = reinterpret_cast<const uint4*>(p.view_table);
    uint4* dst = reinterpret_cast<uint4*>(table);
    for (int i = tid; i < p.n_entries * ENTW / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = tid; i < p.W * p.H; i += blockDim.x) cidx[i] = p.cell_index[i];
  }
#if MG_ABL != 5   // (5: timing-only build without the LDS fills, to size the prologue)
  for (int i = tid; i < p.PW * p.PH; i += blockDim.x) map[i] = p.map_pad[i];
  for (int i = tid; i < p.W * p.H; i += blockDim.x) dist[i] = p.dist_lut[i];
#endif
  __syncthreads();
  // The first chunk's inputs are retired here, on every path and before anything else is requested: the LDS fills above
  // have long covered their latency.  (A wait under `if (c < n_chunks)` made hipcc copy the registers into the asm
  // operands before the wait and back after it, i.e. read them while the lint's CFG still had them in flight.)
  wait_chunk<0>(cur);

  uint32_t* stage = stage_all + wv * (kWave * ROWB / 4);
  uint32_t err = 0;
  // Vector-memory operations an iteration issues AFTER its prefetch, every one of them for at
  // least one lane of the wave (each chunk has an active lane): obs stores, then env_state and
  // dir_done, and in a step the A reward stores.  A lower bound is all wait_chunk needs.
#if MG_ABL == 0 || MG_ABL >= 3
  constexpr int kObs = A * ((4 * ROWB + kWave - 1) / kWave);
#else
  constexpr int kObs = 0;
#endif
  constexpr int kStep = (kObs + 2 + A) < 63 ? (kObs + 2 + A) : 63, kReset = (kObs + 2) < 63 ? (kObs + 2) : 63;
  constexpr bool kCanUpfront = A == 3 && V == 5;    // the benchmark shape only: four more inlined copies of do_chunk
  if (FAST && kCanUpfront && p.n_chunks <= cstride * kUpfront) {
    // Launch-resident batch (at most kUpfront chunks per wave): fetch ALL inputs first, then only
    // store.  The whole chip then runs one read burst followed by an uninterrupted write stream,
    // which the memory system serves faster than reads interleaved with writes.
    // (named variables, not an array: whether a loop over an array unrolls depends on the size of do_chunk, and an
    // array indexed at run time lives in scratch, where it would be stored while its loads are still in flight)
    static_assert(kUpfront == 4, "four explicit slots below");
    // Chunk of this wave in window k (a window = one chunk for every wave of the launch).  A batch that is not a whole
    // number of windows ends in a partial one; dealt out like the others it would leave the workgroups at the end of the
    // grid idle for that window (917,504 envs = 3.5 windows: half the CUs stop after three chunks).  The partial window
    // is therefore dealt out transposed — consecutive chunks to consecutive workgroups — so that every CU gets its share.
    const int64_t c0 = c;
    auto chunk_of = [&](int k) -> int64_t {
      const int64_t base = (int64_t)k * cstride;
      const bool partial = k > 0 && base + cstride > p.n_chunks;
      const int64_t ck = base + (partial ? (int64_t)wv * gridDim.x + blockIdx.x : c0);
      return ck < p.n_chunks ? ck : (int64_t)-1;
    };
    const int64_t c1 = chunk_of(1), c2 = chunk_of(2), c3 = chunk_of(3);
    const bool h1 = c1 >= 0, h2 = c2 >= 0, h3 = c3 >= 0;
    ChunkIn in1, in2, in3;
    zero_chunk(in1); zero_chunk(in2); zero_chunk(in3);
    if (h1) load_chunk<A, FAST>(p, c1 * kWave, lane, in1);
    if (h2) load_chunk<A, FAST>(p, c2 * kWave, lane, in2);
    if (h3) load_chunk<A, FAST>(p, c3 * kWave, lane, in3);
    wait_chunk<0>(in1); wait_chunk<0>(in2); wait_chunk<0>(in3);
    if (c < p.n_chunks) do_chunk<A, V, TAB, FAST>(p, map, dist, table, cidx, stage, lane, c * kWave, cur, err);
    if (h1) do_chunk<A, V, TAB, FAST>(p, map, dist, table, cidx, stage, lane, c1 * kWave, in1, err);
    if (h2) do_chunk<A, V, TAB, FAST>(p, map, dist, table, cidx, stage, lane, c2 * kWave, in2, err);
    if (h3) do_chunk<A, V, TAB, FAST>(p, map, dist, table, cidx, stage, lane, c3 * kWave, in3, err);
  } else {
    for (; c < p.n_chunks; c += cstride) {
      const int64_t cn = c + cstride;
      const bool more = cn < p.n_chunks;          // wave-uniform
      if (more) load_chunk<A, FAST>(p, cn * kWave, lane, nxt);
      do_chunk<A, V, TAB, FAST>(p, map, dist, table, cidx, stage, lane, c * kWave, cur, err);
      if (more) {
        if (FAST || p.op == OP_STEP) wait_chunk<kStep>(nxt); else wait_chunk<kReset>(nxt);
        cur = nxt;
      }
    }
  }
  if (err) atomicOr(p.err, err);
}

// Fused K-step synthetic rollout (mg_rollout_random): p.n_roll consecutive free-running steps with on-device random
// actions — exactly what p.n_roll calls of mg_step_random compute — in ONE launch.  A wave loads its chunk's env_state
// once, keeps it in registers over the K steps and writes it back once; step k's obs / rewards / dir-done / actions go
// to slot k of the caller's trajectory buffers (write-once data: non-temporal stores), the last step's also to the
// handle's own buffers.  What this removes from a step: the launch ramp and drain (~8 us per launch, the whole cost of
// a step below ~100k envs) and the env_state / action traffic.
// STD = the switches of the configuration the reference trains with — v8 shaping, auto-reset, not competitive, drawn
// spawns, float64 rewards, opaque walls — are compile-time constants (launch() picks it when the handle matches): the
// branches and the scalar registers that carry them go (round 3: 6.05 -> 5.69 us per step at 4096 envs, 7.00 -> 6.71 at
// 131072; nothing at 1M envs, where the kernel waits on its stores).
template <int A, int V, bool TAB, bool STD = false>
__global__ __launch_bounds__(kMaxBlock) void mg_rollout_kernel(const Params p) {
  constexpr int ROWB = ViewGeom<V>::ROWB, ENTW = ViewGeom<V>::ENTW;
  extern __shared__ __align__(16) uint8_t smem[];
  const int nw = blockDim.x >> 6;
  uint32_t* stage_all = reinterpret_cast<uint32_t*>(smem);
  uint32_t* table = reinterpret_cast<uint32_t*>(smem + (size_t)nw * kWave * ROWB);
  double* dist = reinterpret_cast<double*>(table + (TAB ? p.n_entries * ENTW : 0));
  uint32_t* map = reinterpret_cast<uint32_t*>(dist + p.W * p.H);
  uint16_t* cidx = reinterpret_cast<uint16_t*>(map + p.PW * p.PH);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  if (TAB) {
    const uint4* src = reinterpret_cast<const uint4*>(p.view_table);
    uint4* dst = reinterpret_cast<uint4*>(table);
    for (int i = tid; i < p.n_entries * ENTW / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = tid; i < p.W * p.H; i += blockDim.x) cidx[i] = p.cell_index[i];
  }
  for (int i = tid; i < p.PW * p.PH; i += blockDim.x) map[i] = p.map_pad[i];
  for (int i = tid; i < p.W * p.H; i += blockDim.x) dist[i] = p.dist_lut[i];
  __syncthreads();

  uint32_t* stage = stage_all + wv * (kWave * ROWB / 4);
  uint32_t err = 0;
  const int64_t cstride = (int64_t)gridDim.x * nw;
  const int64_t slot_obs = (int64_t)A * p.stride * ROWB, slot_rew = (int64_t)A * p.stride;
  const int64_t slot_dd = p.stride * p.dd_bytes, slot_act = p.stride * kSlotWords<A>;
  Params q = p;
  if constexpr (STD) {
    q.shaping = MG_SHAPING_V8; q.auto_reset = 1; q.competitive = 0; q.fixed_mode = MG_FIXED_OFF; q.reward_f32 = 0; q.see_through = 0;
  }
  // A wave owns up to four chunks (one per "window" of cstride chunks, like the launch-resident step kernel) and takes
  // them through the K steps TOGETHER — step k of all its chunks, then step k + 1 — so that the whole chip writes one
  // trajectory slot at a time, as one sweeping front of contiguous regions.  (One chunk per wave for all K steps, then
  // the next chunk, scatters the writes over K slots x all agents and reaches only ~4.5 TB/s.)  The state of a chunk
  // is six registers; the four sets rotate through set 0, so the big step body exists once.
  // (Round 3: the rotation's struct selects keep the sets in 40 bytes of scratch per lane, and the s_waitcnt a scratch
  //  load needs drains the chunk's obs stores each step.  Sets held in scalars with an unconditional four-way rotation
  //  removed the scratch traffic and were SLOWER — 7.18 -> 7.86 us/step at 131072 x K32, 58.9 -> 60.2 at 1M x K16 — so
  //  the form below stays.  What bounds the rollout is instruction issue: its one step body has 1722 VALU instructions
  //  against 1485 per chunk in the step kernel, 259 of them v_readlane reloads of spilled SGPRs against 62.)
  struct St { pd_t<A> pd; uint32_t shw[A]; uint32_t st; };
  constexpr int NJ = 4;
  for (int64_t g0 = (int64_t)blockIdx.x * nw + wv; g0 < p.n_chunks; g0 += cstride * NJ) {
    St s0, s1, s2, s3;
    int nj = 0;
    auto load_state = [&](int jj, St& s) {
      const int64_t c = g0 + jj * cstride;
      ChunkIn in;
      zero_chunk(in);
      if (c < p.n_chunks) {                      // wave-uniform
        nj = jj + 1;
        const int64_t n = c * kWave + lane;
        if (A == 3) {
          const uint4 w = reinterpret_cast<const uint4*>(p.env_state)[n];
          in.q = u32x4{w.x, w.y, w.z, w.w};
        } else {
          const uint32_t* sp = p.env_state + n * (A + 1);
          in.w0 = sp[0];
          if (A >= 1) in.w1 = sp[1];
          if (A >= 2) in.w2 = sp[2];
          if (A >= 3) in.w3 = sp[3];
          if (A >= 4) in.w4 = sp[4];
          if (A >= 5) in.w5 = sp[5];
        }
      }
      uint32_t actw;
      unpack_chunk<A>(p, in, s.pd, s.shw, s.st, actw);
    };
    load_state(0, s0); load_state(1, s1); load_state(2, s2); load_state(3, s3);
#pragma unroll 1
    for (int k = 0; k < p.n_roll; ++k) {
      const bool last = k == p.n_roll - 1;
      q.t = p.t + (uint64_t)k;
      q.obs_image = p.traj_obs + k * slot_obs;
      q.reward = reinterpret_cast<double*>(reinterpret_cast<char*>(p.traj_reward) + k * slot_rew * (!STD && p.reward_f32 ? 4 : 8));
      q.dir_done = p.traj_dd + k * slot_dd;
      q.actions_out = p.traj_actions ? p.traj_actions + k * slot_act : nullptr;
      q.actions = p.actions ? p.actions + k * slot_act : nullptr;
#pragma unroll 1
      for (int jj = 0; jj < nj; ++jj) {
        step_chunk<A, V, TAB, true, true>(q, map, dist, table, cidx, stage, lane, (g0 + jj * cstride) * kWave, s0.pd, s0.shw, s0.st,
                                          0u, last, err, last ? p.obs_image : nullptr, last ? p.reward : nullptr,
                                          last ? p.dir_done : nullptr);
        if (nj > 1) {                            // rotate the first nj sets: set 0 is always the one worked on
          const St t = s0;
          s0 = s1;
          s1 = nj > 2 ? s2 : t;
          s2 = nj > 3 ? s3 : t;
          s3 = t;
        }
      }
    }
  }
  if (err) atomicOr(p.err, err);
}

// The fused rollout for SMALL batches, agents dealt out over waves (the north-star mapping's idea, where it pays): a batch of a few
// thousand envs leaves most of the chip idle and a step's time is the latency of one wave working through one chunk, 60 % of it
// the three views and their staging.  Here A waves share a chunk (step_chunk<..., SPLIT>): each keeps the chunk's whole state in
// its registers and runs the whole (serial, cheap) action phase, so they stay in step without ever talking to each other — same
// Philox draws, same arithmetic — and each produces one agent's shaper, reward, view and image.  launch() picks it while a CU's
// share of chunks times A fits one workgroup (mg_handle::roll_split_fn); beyond that the plain rollout kernel is the faster one,
// because the action phase done A times over is then paid in throughput.  Lives in mgrid_dynroll.hip's code object.
template <int A, int V, bool TAB, bool STD>
__global__ __launch_bounds__(kMaxBlock) void mg_rollout_split_kernel(const Params p) {
  constexpr int ROWB = ViewGeom<V>::ROWB, ENTW = ViewGeom<V>::ENTW;
  extern __shared__ __align__(16) uint8_t smem[];
  const int nw = blockDim.x >> 6;
  uint32_t* stage_all = reinterpret_cast<uint32_t*>(smem);
  uint32_t* table = reinterpret_cast<uint32_t*>(smem + (size_t)nw * kWave * ROWB);
  double* dist = reinterpret_cast<double*>(table + (TAB ? p.n_entries * ENTW : 0));
  uint32_t* map = reinterpret_cast<uint32_t*>(dist + p.W * p.H);
  uint16_t* cidx = reinterpret_cast<uint16_t*>(map + p.PW * p.PH);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  if (TAB) {
    const uint4* src = reinterpret_cast<const uint4*>(p.view_table);
    uint4* dst = reinterpret_cast<uint4*>(table);
    for (int i = tid; i < p.n_entries * ENTW / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = tid; i < p.W * p.H; i += blockDim.x) cidx[i] = p.cell_index[i];
  }
  for (int i = tid; i < p.PW * p.PH; i += blockDim.x) map[i] = p.map_pad[i];
  for (int i = tid; i < p.W * p.H; i += blockDim.x) dist[i] = p.dist_lut[i];
  __syncthreads();

  uint32_t* stage = stage_all + wv * (kWave * ROWB / 4);
  uint32_t err = 0;
  const int64_t slot_obs = (int64_t)A * p.stride * ROWB, slot_rew = (int64_t)A * p.stride;
  const int64_t slot_dd = p.stride * p.dd_bytes, slot_act = p.stride * kSlotWords<A>;
  Params q = p;
  if constexpr (STD) {
    q.shaping = MG_SHAPING_V8; q.auto_reset = 1; q.competitive = 0; q.fixed_mode = MG_FIXED_OFF; q.reward_f32 = 0; q.see_through = 0;
  }
  // the waves of a chunk — p.roll_split of them, two to A; wave `me` takes agents me, me + parts, ... — sit in one workgroup
  // (launch(): nw is a multiple of parts) and meet at a barrier between reading the chunk's state and the K steps, at whose end
  // they write it back: no wave can read what another has already advanced.  The trip count is the same for all waves of the
  // workgroup; a wave without a chunk only keeps the barriers company.
  const int parts = (int)p.roll_split, cpb = nw / parts, me = wv % parts;
  uint32_t mine = 0u;
  for (int a = me; a < A; a += parts) mine |= 1u << a;
  const int64_t n_groups = (p.n_chunks + cpb - 1) / cpb;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t c = g * cpb + wv / parts;
    const bool valid = c < p.n_chunks;
    ChunkIn in;
    zero_chunk(in);
    const int64_t n = c * kWave + lane;
    if (valid) {
      if (A == 3) {
        const uint4 w = reinterpret_cast<const uint4*>(p.env_state)[n];
        in.q = u32x4{w.x, w.y, w.z, w.w};
      } else {
        const uint32_t* sp = p.env_state + n * (A + 1);
        in.w0 = sp[0];
        if (A >= 1) in.w1 = sp[1];
        if (A >= 2) in.w2 = sp[2];
        if (A >= 3) in.w3 = sp[3];
        if (A >= 4) in.w4 = sp[4];
        if (A >= 5) in.w5 = sp[5];
      }
    }
    __syncthreads();
    if (!valid) continue;
    pd_t<A> pd;
    uint32_t shw[A], st, actw;
    unpack_chunk<A>(p, in, pd, shw, st, actw);
#pragma unroll 1
    for (int k = 0; k < p.n_roll; ++k) {
      const bool last = k == p.n_roll - 1;
      q.t = p.t + (uint64_t)k;
      q.obs_image = p.traj_obs + k * slot_obs;
      q.reward = reinterpret_cast<double*>(reinterpret_cast<char*>(p.traj_reward) + k * slot_rew * (!STD && p.reward_f32 ? 4 : 8));
      q.dir_done = p.traj_dd + k * slot_dd;
      q.actions_out = (p.traj_actions && me == 0) ? p.traj_actions + k * slot_act : nullptr;
      q.actions = p.actions ? p.actions + k * slot_act : nullptr;
      step_chunk<A, V, TAB, true, true, true>(q, map, dist, table, cidx, stage, lane, c * kWave, pd, shw, st, 0u, last, err,
                                              last ? p.obs_image : nullptr, last ? p.reward : nullptr, last ? p.dir_done : nullptr, mine);
    }
  }
  if (err) atomicOr(p.err, err);
}

#include "mg_dyn.h"

// The instantiations of the mutable-grid kernels, one to five agents for view 7, one to four for view 5 and the fully observed
// kinds (no views; the V parameter is unused), with 16 or 64 prefetched grid rows.  X(A, V, FULL, PRE).
// MG_MIN_BUILD (tools/gpu.sh build ... -DMG_MIN_BUILD): only what tools/kbench times — quick A/B builds, never the product
#ifdef MG_MIN_BUILD
#define MG_DYN_LIST(X) X(3, 7, false, 16) X(2, 7, false, 16) X(3, 5, false, 64) X(3, 5, true, 16) X(3, 5, true, 64)
#else
#define MG_DYN_LIST(X) \
    X(1, 7, false, 16) X(2, 7, false, 16) X(3, 7, false, 16) X(4, 7, false, 16) X(5, 7, false, 16) \
    X(1, 5, false, 16) X(2, 5, false, 16) X(3, 5, false, 16) X(4, 5, false, 16) \
    X(1, 5, false, 64) X(2, 5, false, 64) X(3, 5, false, 64) \
    X(1, 5, true, 16) X(2, 5, true, 16) X(3, 5, true, 16) X(4, 5, true, 16) \
    X(1, 5, true, 64) X(2, 5, true, 64) X(3, 5, true, 64)
#endif

#ifdef MG_TU_DYNROLL
// ---- second translation unit (mgrid_dynroll.hip): the mutable-grid kinds' fused rollout and free-running step kernels, the
//      shared-map rollout with the agents dealt out over waves, and their lookup functions — nothing else ----
struct DynRollEntry { int A, V; bool full; int pre; const void* fn[2]; const void* step_free[2]; };
#define MG_DYN_ROLL(A, V, FULL, PRE) {A, FULL ? 0 : V, FULL, PRE, {(const void*)mg_dyn_rollout_kernel<A, V, FULL, PRE, false>, \
                                      FULL ? (const void*)nullptr : (const void*)mg_dyn_rollout_kernel<A, V, FULL, PRE, !FULL>}, \
                                      {(const void*)mg_dyn_kernel<A, V, FULL, PRE, false, true>, \
                                      FULL ? (const void*)nullptr : (const void*)mg_dyn_kernel<A, V, FULL, PRE, !FULL, true>}},
const DynRollEntry kDynRoll[] = { MG_DYN_LIST(MG_DYN_ROLL) };

// shared-map fused rollout with the agents dealt out over waves (mg_rollout_split_kernel): [table path][standard configuration]
struct SplitEntry { int A, V; const void* fn[2][2]; };
#define MG_SPLIT_TAB(A, V) {A, V, {{(const void*)mg_rollout_split_kernel<A, V, false, false>, (const void*)mg_rollout_split_kernel<A, V, false, true>}, \
                                   {(const void*)mg_rollout_split_kernel<A, V, true, false>, (const void*)mg_rollout_split_kernel<A, V, true, true>}}},
#define MG_SPLIT_GEN(A, V) {A, V, {{(const void*)mg_rollout_split_kernel<A, V, false, false>, (const void*)mg_rollout_split_kernel<A, V, false, true>}, \
                                   {nullptr, nullptr}}},
const SplitEntry kSplit[] = {
#ifdef MG_MIN_BUILD
    MG_SPLIT_TAB(3, 5)
#else
    MG_SPLIT_TAB(2, 5) MG_SPLIT_TAB(3, 5) MG_SPLIT_TAB(4, 5) MG_SPLIT_TAB(5, 5) MG_SPLIT_TAB(3, 3) MG_SPLIT_GEN(2, 7) MG_SPLIT_GEN(3, 7)
#endif
};
}  // namespace

extern "C" __attribute__((visibility("hidden"))) const void* mg_rollsplit_kernel(int A, int V, int table, int std_cfg) {
  for (const SplitEntry& e : kSplit)
    if (e.A == A && e.V == V) return e.fn[table ? 1 : 0][std_cfg ? 1 : 0];
  return nullptr;
}

// which: 0 = the fused rollout kernel, 1 = the free-running step kernel (mg_dyn_kernel<..., FREE>)
extern "C" __attribute__((visibility("hidden"))) const void* mg_dynroll_kernel(int A, int V, int full, int pre, int dec, int which) {
  for (const DynRollEntry& e : kDynRoll)
    if (e.A == A && e.V == V && e.full == (full != 0) && e.pre == pre) return (which ? e.step_free : e.fn)[dec ? 1 : 0];
  return nullptr;
}
#else
// (defined in mgrid_dynroll.hip's translation unit; C linkage, hidden: not part of the library's ABI)
extern "C" __attribute__((visibility("hidden"))) const void* mg_dynroll_kernel(int A, int V, int full, int pre, int dec, int which);
extern "C" __attribute__((visibility("hidden"))) const void* mg_rollsplit_kernel(int A, int V, int table, int std_cfg);

// ---------------------------------------------------------------------------------------------
// mg_step_noalloc: the per-GPU scratch its kernels use instead of a handle's hipMalloc'ed tables.
// Layout: [0, 256) the error word (+ padding), [256, 256 + 32 KB) the goal-distance LUT (W*H f64, W, H <= 64),
// then the padded map ((W + 2(V-1)) * (H + 2(V-1)) u32, V <= 7: at most 23,104 B).  Host code never touches the
// array by name: it goes through the per-device address hipGetSymbolAddress returned (noalloc_device).
// ---------------------------------------------------------------------------------------------
constexpr size_t kScratchBytes = 64 * 1024;
constexpr size_t kScratchDist = 256, kScratchMap = 256 + 64 * 64 * 8;
static_assert(kScratchMap + (64 + 12) * (64 + 12) * 4 <= kScratchBytes, "scratch holds the largest grid the kernels take");
static __device__ __align__(256) unsigned char g_scratch[kScratchBytes];

// One workgroup turns the caller's layout (device uint8[H*W*3], the cells mg_set_static_map takes from the host) into
// the padded map and the distance LUT, exactly as mg_set_static_map builds them on the host.  Unknown or mutable cell
// types cannot be refused from the device: they are flagged (MG_ERR_BAD_MAP) and treated as walls.
__global__ __launch_bounds__(256) void noalloc_prep(const uint8_t* cells, int W, int H, int V, int A, uint32_t* err,
                                                    double* dist, uint32_t* map) {
  __shared__ unsigned int goal_key, n_free, bad;
  const int pad = V - 1, PW = W + 2 * pad, PH = H + 2 * pad;
  if (threadIdx.x == 0) { goal_key = 0xFFFFFFFFu; n_free = 0; bad = 0; }
  __syncthreads();
  const uint32_t wall = (uint32_t)MG_T_WALL | (5u << 8) | F_OPAQUE;   // Grid.slice's out-of-bounds Wall() (multigrid.py:315)
  for (int i = threadIdx.x; i < PW * PH; i += blockDim.x) {
    const int x = i % PW - pad, y = i / PW - pad;
    uint32_t w = wall;
    if (x >= 0 && y >= 0 && x < W && y < H) {
      const uint8_t* c = cells + ((size_t)y * W + x) * 3;
      w = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
      switch (c[0]) {
        case MG_T_UNSEEN: case MG_T_EMPTY: w = (uint32_t)MG_T_EMPTY | F_FREE; atomicAdd(&n_free, 1u); break;
        case MG_T_WALL: w |= F_OPAQUE; break;
        case MG_T_GOAL: w |= F_KILL | F_GOAL; atomicMin(&goal_key, (unsigned int)(x * H + y)); break;   // first goal, x-major (v8:159-169)
        case MG_T_LAVA: if (c[1] == 5) w = (uint32_t)MG_T_WALL | (5u << 8); w |= F_KILL; break;         // grey lava = LavaWall
        default: w = wall; atomicOr(&bad, 1u); break;
      }
    }
    map[i] = w;
  }
  __syncthreads();
  const int gx = goal_key == 0xFFFFFFFFu ? 13 : (int)(goal_key / (unsigned int)H);
  const int gy = goal_key == 0xFFFFFFFFu ? 13 : (int)(goal_key % (unsigned int)H);
  for (int i = threadIdx.x; i < W * H; i += blockDim.x) {
    const long dx = i % W - gx, dy = i / W - gy;
    dist[i] = __dsqrt_rn((double)(dx * dx + dy * dy));   // np.linalg.norm (v8:192); checked against the host's sqrt for
  }                                                       // every reachable argument by tests (mg_debug_device_sqrt)
  if (threadIdx.x == 0 && (bad || n_free < (unsigned int)A)) atomicOr(err, MG_ERR_BAD_MAP);
}

__global__ void sqrt_probe(int n, double* out) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += gridDim.x * blockDim.x) out[k] = __dsqrt_rn((double)k);
}

// mg_set_maps: the caller's per-env layouts (device uint8[n_envs][H*W*3] cell triples) -> cell codes in the blocked layout of
// the per-env grids, plus every env's own shaper goal (get_goal_position, v8_robust_ppo.py:159-169: first goal scanning
// x-major, (13,13) if none) into aux word 0.  One thread per env: a one-off import, not a hot path.
__global__ void import_maps_kernel(const uint8_t* cells, int W, int H, int GB, int A, int64_t n_envs, uint8_t* grid0, uint32_t* aux,
                                   uint32_t* err) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_envs) return;
  const int ncell = W * H;
  const uint8_t* c = cells + n * ncell * 3;
  uint8_t* g = grid0 + (n >> 6) * (int64_t)GB * (kWave * 16);      // this env's chunk block; grid_byte() places (cell, lane)
  int n_free = 0;
  bool bad = false;
  for (int ci = 0; ci < ncell; ++ci) {
    const uint32_t t = c[ci * 3], col = c[ci * 3 + 1] & 7u, st = c[ci * 3 + 2];
    uint32_t k = K_NONE;
    switch (t) {
      case MG_T_UNSEEN: case MG_T_EMPTY: ++n_free; break;
      case MG_T_WALL: case MG_T_GOAL: case MG_T_LAVA: case MG_T_FLOOR: case MG_T_KEY: case MG_T_BALL: case MG_T_BOX:
        k = code_from_kind(t * 8 + col); break;
      case MG_T_DOOR: if (st > 2) { bad = true; k = K_WALL; } else k = code_from_kind(128 + col * 4 + st); break;
      default: bad = true; k = K_WALL; break;
    }
    g[grid_byte(ci, (int)(n & 63))] = (uint8_t)k;
  }
  uint32_t goal = 13u | (13u << 6);
  bool found = false;
  for (int i = 0; i < W && !found; ++i)
    for (int j = 0; j < H; ++j)
      if (c[(j * W + i) * 3] == MG_T_GOAL) { goal = (uint32_t)i | ((uint32_t)j << 6); found = true; break; }
  aux[n * 4] = goal | 0xFFFF0000u;
  aux[n * 4 + 1] = aux[n * 4 + 2] = 0xFFFFFFFFu;
  aux[n * 4 + 3] = 0u;
  if (bad || n_free < A) atomicOr(err, MG_ERR_BAD_MAP);
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

// [0] general / [1] FAST instantiation of the generic-path and table-path kernels
// roll[0] / roll[1]: the fused rollout on the generic / table path; roll_std: the same with the standard configuration's switches folded
struct Variant { int A, V; kernel_fn generic[2], table[2]; int entw; kernel_fn roll[2], roll_std[2]; };
#define MG_VARIANT_TAB(A, V) {A, V, {mg_kernel<A, V, false, false>, mg_kernel<A, V, false, true>}, \
                              {mg_kernel<A, V, true, false>, mg_kernel<A, V, true, true>}, ViewGeom<V>::ENTW, \
                              {mg_rollout_kernel<A, V, false>, mg_rollout_kernel<A, V, true>}, \
                              {mg_rollout_kernel<A, V, false, true>, mg_rollout_kernel<A, V, true, true>}}
#define MG_VARIANT_GEN(A, V) {A, V, {mg_kernel<A, V, false, false>, mg_kernel<A, V, false, true>}, {nullptr, nullptr}, 0, \
                              {mg_rollout_kernel<A, V, false>, nullptr}, {mg_rollout_kernel<A, V, false, true>, nullptr}}
// MG_MIN_BUILD (tools/gpu.sh build ... -DMG_MIN_BUILD): only what tools/kbench times — quick A/B builds, never the product
const Variant kVariants[] = {
#ifdef MG_MIN_BUILD
    MG_VARIANT_TAB(3, 5),
#else
    MG_VARIANT_TAB(1, 5), MG_VARIANT_TAB(2, 5), MG_VARIANT_TAB(3, 5), MG_VARIANT_TAB(4, 5), MG_VARIANT_TAB(5, 5), MG_VARIANT_TAB(3, 3),
    MG_VARIANT_GEN(1, 7), MG_VARIANT_GEN(2, 7), MG_VARIANT_GEN(3, 7),
#endif
};

// mutable-grid kernel instantiations
// fn[0] decodes a view cell through the LDS table, fn[1] arithmetically (DEC, mg_dyn.h); the fully observed kinds have no views.
// The fused rollout kernels of the same instantiations (mg_dyn_rollout_kernel) live in a code object of their own
// (mgrid_dynroll.hip, looked up through mg_dynroll_kernel at mg_create): with them in this one — 38 more kernels, 4.3 MB of code
// instead of 3.3 — the shared-map headline kernel, whose ISA and descriptor were bit-identical, ran 1.3 % slower (50.83 against
// 50.15 us at 1M envs, interleaved in one process, profiles/r3_product_vs_session_start_static_ab.txt).
struct DynVariant { int A, V; bool full; int pre; kernel_fn fn[2]; };
#define MG_DYN(A, V, FULL, PRE) {A, FULL ? 0 : V, FULL, PRE, {mg_dyn_kernel<A, V, FULL, PRE, false>, \
                                 FULL ? (kernel_fn) nullptr : (kernel_fn)mg_dyn_kernel<A, V, FULL, PRE, !FULL>}},
const DynVariant kDynVariants[] = { MG_DYN_LIST(MG_DYN) };

// Host twin of static_view<V>: the static view of (x, y, dir) as a table entry
// (NW stream dwords of 3-byte codes with visibility applied, then the visibility bit mask).
template <int V>
void host_view_entry(const std::vector<uint32_t>& map, int PW, int pad, bool see_through, int x, int y, int d,
                     uint32_t* out) {
  constexpr int ROWB = ViewGeom<V>::ROWB, NW = ViewGeom<V>::NW, ENTW = ViewGeom<V>::ENTW;
  const int fx = (d == 0) - (d == 2), fy = (d == 1) - (d == 3), rx = -fy, ry = fx;
  uint32_t code[V * V], opq[V], vis[V];
  for (int j = 0; j < V; ++j) opq[j] = 0;
  for (int vx = 0; vx < V; ++vx)
    for (int vy = 0; vy < V; ++vy) {
      const int wx = x + fx * (V - 1 - vy) + rx * (vx - V / 2), wy = y + fy * (V - 1 - vy) + ry * (vx - V / 2);
      const uint32_t w = map[(size_t)(wy + pad) * PW + (wx + pad)];
      code[vx * V + vy] = w & 0xFFFFFFu;
      opq[vy] |= (w >> 31) << vx;
    }
  if (see_through) for (int j = 0; j < V; ++j) vis[j] = (1u << V) - 1u;
  else mg_vis_rows<V>(opq, vis);
  code[(V / 2) * V + (V - 1)] = (uint32_t)MG_T_EMPTY;
  uint8_t bytes[ENTW * 4];
  memset(bytes, 0, sizeof(bytes));
  uint32_t vm = 0;
  for (int vx = 0; vx < V; ++vx)
    for (int vy = 0; vy < V; ++vy) {
      const int c = vx * V + vy;
      const uint32_t seen = (vis[vy] >> vx) & 1u, v = seen ? code[c] : 0u;
      vm |= seen << (vy * V + vx);
      bytes[3 * c] = (uint8_t)v; bytes[3 * c + 1] = (uint8_t)(v >> 8); bytes[3 * c + 2] = (uint8_t)(v >> 16);
    }
  static_assert(ROWB <= NW * 4, "");
  memcpy(out, bytes, ENTW * 4);
  out[NW] = vm;
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


// Episode statistics (mg_enable_episode_stats): what run_episode keeps beside the trajectories (v8_robust_ppo.py:234-235,
// 246-248, 266, 273) — per-agent return of the episode in progress, its goal touches (rewards >= threshold), its length — and,
// when the step reports done, the finished episode's figures and the episode count.  One thread per env; K > 1 walks the K
// steps of a fused rollout's trajectory.  Additions happen per env in step order, so the sums are the reference's floats.
struct EpStats {
  double* ep_return; uint32_t* ep_goal_touches; uint32_t* ep_len;
  double* last_return; uint32_t* last_goal_touches; uint32_t* last_len; uint32_t* ep_count;
};

template <typename R>
__global__ void ep_stats_kernel(EpStats st, const R* __restrict__ reward, const uint8_t* __restrict__ dd, int dd_bytes, int A, int K,
                                int64_t n_envs, int64_t stride, double thr, const uint8_t* __restrict__ reset_mask, int is_reset) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_envs) return;
  if (is_reset) {                                 // env.reset(): the episode in progress is dropped (v8:224-235)
    if (reset_mask && !reset_mask[n]) return;
    for (int a = 0; a < A; ++a) st.ep_return[a * stride + n] = 0.0;
    st.ep_goal_touches[n] = 0; st.ep_len[n] = 0;
    return;
  }
  constexpr int kMaxA = 5;                         // the largest n_agents any kernel here is instantiated for
  double ret[kMaxA];
#pragma unroll
  for (int a = 0; a < kMaxA; ++a) ret[a] = a < A ? st.ep_return[a * stride + n] : 0.0;
  uint32_t touches = st.ep_goal_touches[n], len = st.ep_len[n], count = st.ep_count[n];
  bool finished = false;
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int a = 0; a < kMaxA; ++a)
      if (a < A) {
        const double r = (double)reward[((int64_t)k * A + a) * stride + n];
        ret[a] += r;
        touches += (r >= thr) ? 1u : 0u;
      }
    len += 1;
    if (dd[((int64_t)k * stride + n) * dd_bytes + A]) {
#pragma unroll
      for (int a = 0; a < kMaxA; ++a)
        if (a < A) { st.last_return[a * stride + n] = ret[a]; ret[a] = 0.0; }
      st.last_goal_touches[n] = touches; st.last_len[n] = len;
      count += 1; finished = true;
      touches = 0; len = 0;
    }
  }
#pragma unroll
  for (int a = 0; a < kMaxA; ++a)
    if (a < A) st.ep_return[a * stride + n] = ret[a];
  st.ep_goal_touches[n] = touches; st.ep_len[n] = len;
  if (finished) st.ep_count[n] = count;
}

}  // namespace

struct mg_handle {
  mg_config cfg;
  int device;
  hipStream_t stream;
  bool own_stream;
  const Variant* var;
  bool use_table;
  int n_entries;
  int64_t stride, n_chunks;
  int PW, PH, n_cu, dd_bytes;
  bool have_map, have_reset;
  uint64_t t;
  uint64_t reset_epoch;  // explicit mg_reset* calls so far
  int gx, gy;
  // device memory
  uint32_t* map_pad;
  double* dist_lut;
  uint32_t* view_table;
  uint16_t* cell_index;
  uint32_t* err;
  // mutable-grid path (DoorKey, layouts with objects)
  bool dynamic;
  const DynVariant* dvar;
  int GW;
  uint32_t* base_grid;
  uint32_t* grid0;       // per-env reset templates (mg_set_maps), same blocked layout as grid
  bool per_env_maps;
  uint32_t* carry;
  uint32_t* grid;
  uint32_t* aux;
  uint8_t* obs_full;
  bool full_obs;
  int dyn_waves_cu2[2];  // waves of the mutable-grid kernel (table-decoding / arithmetic) a CU can hold by registers
  int dyn_roll_waves_cu2[2];   // the same for the fused rollout kernels
  const void* dyn_roll_fn[2];  // those kernels (mgrid_dynroll.hip's code object), null where there is none
  const void* roll_split_fn[2][2];   // mg_rollout_split_kernel [table path][standard configuration], null where there is none
  const void* dyn_free_fn[2];  // mg_dyn_kernel<..., FREE>: the free-running step (same code object)
  int dyn_free_waves_cu2[2];
  bool dyn_dec;          // this layout family's views decode their cell codes arithmetically (DEC, mg_dyn.h)
  int fixed_mode; uint64_t fixed_pd; uint32_t fixed_pd_hi;       // mg_set_fixed_spawn
  std::vector<uint8_t> free_cells;         // host copy: cell (y*W + x) holds nothing (placement checks)
  int LW, LH;            // dims of the mutable-grid kernel's relative distance LUT
  bool ep_on; double ep_thr; EpStats ep;   // mg_enable_episode_stats
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

size_t lds_fixed_bytes(const mg_handle* h, bool table) {
  const mg_config& c = h->cfg;
  size_t b = (size_t)c.width * c.height * 8 + (size_t)h->PW * h->PH * 4;
  if (table) b += (size_t)h->n_entries * h->var->entw * 4 + (size_t)c.width * c.height * 2;
  return (b + 15) / 16 * 16;
}

// Debug / tuning overrides (tests, tools/): an unset or empty variable means "no override".
const char* debug_env(const char* name) {
  const char* e = getenv(name);
  return (e && *e) ? e : nullptr;
}

// Launch geometry: workgroups are persistent and stride over the 64-env chunks.  Waves per
// workgroup are chosen so that (a) the LDS tables + per-wave staging fit 160 KB and (b) a small
// batch still spreads over all CUs.
// Which shared-map kernel a launch runs.  Up to kUpfront chunks per resident wave (1,048,576 envs on 256 CUs) the
// persistent table kernel is fastest: its whole batch is launch-resident and it reads everything before it writes.
// Beyond that the working set no longer lives in the Infinity Cache, and a write stream then runs ~20 % faster when
// short-lived workgroups write their chunks in dispatch order than when persistent waves stride over the buffers
// (tools/wstream.cpp).  Only the generic kernels can be launched that way — they have next to no
// per-workgroup prologue — so a large batch takes them, one chunk per wave, unless the caller forced a path.
bool launch_uses_table(const mg_handle* h) {
  if (!h->use_table) return false;
  if (h->cfg.kernel_path == MG_PATH_TABLE) return true;
  // a step launch pays ~3 us of LDS fill per workgroup for the table, which only pays off from ~64K envs
  // (measured: 32768 envs 10.4 us generic vs 11.0 us table; 131072 envs 14.7 vs 12.5)
  return h->n_chunks >= 1024 && h->n_chunks <= (int64_t)h->n_cu * 16 * kUpfront;
}

// Whether the obs images leave with non-temporal stores (store_vec16): when a launch writes more obs than the 256 MB
// Infinity Cache can usefully hold.  Shared-map kernels: beyond the launch-resident batch, the same boundary as
// launch_uses_table (1.3M envs 67.4 -> 61.8 us, 2M 114.3 -> 96.4, 4M 216.1 -> 192.3, 8M 427 -> 412; at 1M the resident
// table kernel loses: 51.5 -> 57.7 us).  Mutable-grid kernels: from 192 MiB of obs per launch (DoorKey-8x8x3 at 1M envs
// 167.3 -> 148.6 us, Meetup-15x15x3 307.4 -> 291.2; no difference either way below that).  tools/nt_sweep.sh, DESIGN §6.
// The launch-resident table kernel with 192 MiB of obs or more (the 1M-env bench shape: 236 MB) streams every agent's
// images but the last one's: 50.4 us against 51.55 with none and 58.2 with all of them streamed, the same on three
// runs (tools/nt_mask2.sh; streaming one agent's 50.8, another two 50.8 - 51.3).
int launch_streams_obs(const mg_handle* h) {     // bit a = agent a's images
  if (const char* e = debug_env("MG_DEBUG_NT")) return atoi(e);   // tests / tuning experiments only: the mask itself
  const mg_config& c = h->cfg;
  const int64_t per_env = h->full_obs ? (int64_t)c.width * c.height * 3 : (int64_t)c.n_agents * c.view_size * c.view_size * 3;
  const bool big = h->stride * per_env >= ((int64_t)192 << 20);
  if (h->dynamic) return big ? 0xFF : 0;
  if (h->n_chunks > (int64_t)h->n_cu * 16 * kUpfront) return 0xFF;
  return (big && launch_uses_table(h)) ? (1 << (c.n_agents - 1)) - 1 : 0;
}

void launch_geometry(const mg_handle* h, bool table, int* nw_out, unsigned* grid_out, size_t* lds_out) {
  const int rowb = h->cfg.view_size * h->cfg.view_size * 3;
  const size_t fixed = lds_fixed_bytes(h, table), per_wave = (size_t)kWave * rowb;
  int nw = table ? 16 : 4;
  if (const char* e = debug_env("MG_DEBUG_NW")) {   // tuning experiments only (and the geometry up to round 3's end: powers of two)
    nw = atoi(e);
    while (nw > 1 && (fixed + nw * per_wave > (size_t)kLdsLimit || h->n_chunks < (int64_t)h->n_cu * nw)) nw >>= 1;
  } else {
    while (nw > 1 && fixed + nw * per_wave > (size_t)kLdsLimit) nw >>= 1;
    // a small batch: a workgroup per CU with the CU's share of chunks as its waves (any number — 3, 5, 7 — not only powers
    // of two), so that no CU holds two workgroups' fills and no wave a second chunk while others have none
    nw = (int)std::min<int64_t>(nw, std::max<int64_t>(1, (h->n_chunks + h->n_cu - 1) / h->n_cu));
  }
  const size_t lds = fixed + nw * per_wave;
  int per_cu = (int)((size_t)kLdsLimit / lds);
  per_cu = std::max(1, std::min(per_cu, 32 / nw));
  const int64_t blocks = (h->n_chunks + nw - 1) / nw;
  *nw_out = nw;
  // generic kernels: one chunk per wave (a workgroup per nw chunks, dispatched in address order); table kernels: persistent
  *grid_out = table ? (unsigned)std::min<int64_t>(blocks, (int64_t)h->n_cu * per_cu) : (unsigned)blocks;
  if (const char* e = debug_env("MG_DEBUG_GRID"))   // tests only: few workgroups => many chunks per wave
    *grid_out = (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, atoi(e)));
  *lds_out = lds;
}

// Which instantiation of the mutable-grid kernel a launch runs: the arithmetic decode (DEC) when the layout family
// allows it — no doors, or DoorKey's one tracked door — and the kernel exists (the fully observed kinds have no views).
// MG_DEBUG_DEC=0 forces the table decode (tests: both instantiations on the same layouts).
int dyn_dec_index(const mg_handle* h) {
  if (!h->dvar || !h->dvar->fn[1] || !h->dyn_dec) return 0;
  if (const char* e = debug_env("MG_DEBUG_DEC")) return atoi(e) ? 1 : 0;
  return 1;
}

// Mutable-grid launch geometry: 4 waves per workgroup (fewer for small batches), as many workgroups
// per CU as LDS allows; persistent over chunks.
// parts > 1 (the rollout with a chunk's agents dealt out over waves): a chunk takes `parts` waves of ONE workgroup — they meet at a
// barrier between reading the chunk's state and writing it back — so the workgroup is a multiple of that; false if none fits LDS.
// *one_trip: every chunk has its waves resident at once (no workgroup takes a second turn) — what that mapping needs to pay.
bool dyn_geometry(const mg_handle* h, int* nw_out, unsigned* grid_out, size_t* lds_out, int which = 0, int parts = 1, bool* one_trip = nullptr) {   // which: 0 general step, 1 rollout, 2 free step
  const mg_config& c = h->cfg;
  const size_t rowb = h->full_obs ? 0 : (size_t)c.view_size * c.view_size * 3;   // fully observed images are not staged
  const size_t fixed = ((size_t)h->LW * h->LH * 8 + (size_t)c.width * c.height * 4 + 256 * 4 + (size_t)h->GW * 4 + 8 + 15) / 16 * 16;
  const size_t per_wave = (size_t)kWave * rowb + (size_t)(c.width + 2) * (c.height + 2) * kCellB;
  // waves per workgroup (1..8) that keep the most waves resident per CU: at most dyn_waves_cu by registers
  // (queried from the code object at mg_create: 512 VGPRs per SIMD lane / the kernel's count, x 4 SIMDs), and
  // whatever fits 160 KB of LDS.  Ties go to the smaller workgroup; small batches shrink it further so that
  // every CU gets work.
  const int kDynWavesPerCU = (which == 1 ? h->dyn_roll_waves_cu2 : which == 2 ? h->dyn_free_waves_cu2 : h->dyn_waves_cu2)[dyn_dec_index(h)];
  int nw = 0, best = 0;
  for (int cand = parts; cand <= std::max(8, parts); cand += parts) {
    const size_t need = fixed + cand * per_wave;
    if (need > (size_t)kLdsLimit) break;
    const int resident = parts == 1 ? std::min(kDynWavesPerCU, cand * (int)((size_t)kLdsLimit / need))
                                    : cand * std::min(kDynWavesPerCU / cand, (int)((size_t)kLdsLimit / need));   // whole workgroups
    if (resident > best) { best = resident; nw = cand; }
  }
  if (nw == 0) return false;
  if (parts == 1) nw = (int)std::min<int64_t>(nw, std::max<int64_t>(1, (h->n_chunks + h->n_cu - 1) / h->n_cu));   // the CU's share of chunks, any number (launch_geometry)
  else { while (nw > parts && h->n_chunks * parts < (int64_t)h->n_cu * nw) nw -= parts; }
  const size_t lds = fixed + nw * per_wave;
  int per_cu = (int)((size_t)kLdsLimit / lds);
  per_cu = std::max(1, std::min(per_cu, kDynWavesPerCU / nw));
  const int cpb = nw / parts;                       // chunks per workgroup
  const int64_t blocks = (h->n_chunks + cpb - 1) / cpb;
  *nw_out = nw;
  *grid_out = (unsigned)std::min<int64_t>(blocks, (int64_t)h->n_cu * per_cu);
  if (one_trip) *one_trip = blocks <= (int64_t)h->n_cu * per_cu;
  if (const char* e = debug_env("MG_DEBUG_GRID"))   // tuning experiments / tests only
    *grid_out = (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, atoi(e)));
  *lds_out = lds;
  return true;
}

// After a reset / step / rollout launch: fold its rewards and done flags into the episode statistics (same stream).
int launch_ep_stats(mg_handle* h, int op, const uint8_t* mask, int n_roll, const mg_rollout_out* roll) {
  if (!h->ep_on) return MG_OK;
  const mg_config& c = h->cfg;
  const void* rw = n_roll > 0 ? (const void*)roll->reward : (const void*)h->buf.reward;
  const uint8_t* dd = n_roll > 0 ? roll->dir_done : h->buf.dir_done;
  const int K = n_roll > 0 ? n_roll : 1;
  const unsigned blocks = (unsigned)((c.n_envs + 255) / 256);
  if (c.reward_dtype == MG_REWARD_F32)
    hipLaunchKernelGGL(ep_stats_kernel<float>, dim3(blocks), dim3(256), 0, h->stream, h->ep, (const float*)rw, dd, h->dd_bytes,
                       c.n_agents, K, c.n_envs, h->stride, h->ep_thr, mask, op == OP_RESET ? 1 : 0);
  else
    hipLaunchKernelGGL(ep_stats_kernel<double>, dim3(blocks), dim3(256), 0, h->stream, h->ep, (const double*)rw, dd, h->dd_bytes,
                       c.n_agents, K, c.n_envs, h->stride, h->ep_thr, mask, op == OP_RESET ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return MG_OK;
}

// Whether a free-running step of this handle — mg_step / mg_step_random with no injected draws and no mask, on a handle without the
// env_reward output — is launched as a one-step rollout of the agents-over-waves kernels (mg_rollout_split_kernel; on mutable
// grids mg_dyn_rollout_kernel with roll_split) into the handle's own buffers, and with what geometry.  Small batches only: all
// chunks' A waves resident at two per SIMD on a shared map (generic path), at most four waves per CU on mutable grids (from six
// on it no longer pays there).  MG_DEBUG_STEP_SPLIT=0 keeps the step kernels (tests, A/B timing).
bool step_split_plan(const mg_handle* h, const void** fn, int* nw, unsigned* grid, size_t* lds) {
  const mg_config& c = h->cfg;
  const int A = c.n_agents;
  const char* e = debug_env("MG_DEBUG_STEP_SPLIT");
  if (A < 2 || h->buf.env_reward || (e && atoi(e) == 0)) return false;
  if (h->dynamic) {
    const void* f = h->dyn_roll_fn[dyn_dec_index(h)];
    bool one_trip = false;
    if (!f || !h->dyn_free_fn[dyn_dec_index(h)] || h->n_chunks * A > (int64_t)h->n_cu * 4) return false;
    if (!dyn_geometry(h, nw, grid, lds, 1, A, &one_trip) || !one_trip) return false;
    *fn = f;
    return true;
  }
  if (launch_uses_table(h) || h->n_chunks * A > (int64_t)h->n_cu * 8) return false;
  const bool std_cfg = c.shaping_kind == MG_SHAPING_V8 && c.auto_reset && !c.competitive && h->fixed_mode == MG_FIXED_OFF &&
                       c.reward_dtype != MG_REWARD_F32 && !c.see_through_walls && !debug_env("MG_DEBUG_ROLL_GENERAL");
  const void* f = h->roll_split_fn[0][std_cfg ? 1 : 0];
  if (!f) return false;
  const size_t slds = lds_fixed_bytes(h, false) + (size_t)A * kWave * c.view_size * c.view_size * 3;
  const int per_cu = std::max(1, std::min((int)((size_t)kLdsLimit / slds), 32 / A));
  unsigned sgrid = (unsigned)std::min<int64_t>(h->n_chunks, (int64_t)h->n_cu * per_cu);
  if (const char* ge = debug_env("MG_DEBUG_GRID")) sgrid = (unsigned)std::max(1, std::min<int>(sgrid, atoi(ge)));   // tests only
  *fn = f; *nw = A; *grid = sgrid; *lds = slds;
  return true;
}

// dest (mutable-grid step launches only): this launch writes its obs / rewards / dir-done there instead of into the handle's
// own buffers — a slot of a caller's trajectory (mg_rollout on these kinds).  The handle itself is not touched, so a
// concurrent mg_get_buffers never sees a caller's pointers.
int launch(mg_handle* h, int op, const void* actions, const uint8_t* order, const int8_t* respawn,
           const int8_t* reset_spawn, const uint8_t* mask, int random_actions, uint8_t* actions_out,
           const int8_t* objdrop = nullptr, const int8_t* reset_gen = nullptr, const int8_t* scatter = nullptr,
           int n_roll = 0, const mg_rollout_out* roll = nullptr, const mg_rollout_out* dest = nullptr) {
  Params p;
  memset(&p, 0, sizeof(p));
  const mg_config& c = h->cfg;
  p.W = c.width; p.H = c.height; p.max_steps = c.max_steps; p.see_through = c.see_through_walls;
  p.competitive = (c.n_agents == 1) ? 1 : c.competitive;   // multigrid.py:380-381
  p.shaping = c.shaping_kind; p.auto_reset = c.auto_reset; p.dd_bytes = h->dd_bytes;
  p.op = op; p.random_actions = random_actions; p.PW = h->PW; p.PH = h->PH; p.n_entries = h->n_entries;
  p.n_envs = c.n_envs; p.stride = h->stride; p.n_chunks = h->n_chunks;
  // explicit resets draw under their own counter word — bit 63 + the number of resets so far — so that reset(); reset()
  // redraws (the reference reseeds from the clock, cluttered.py:55-63) and shares nothing with the auto-reset inside step t
  p.gid_base = (uint64_t)c.env_id_base; p.t = op == OP_RESET ? ((1ull << 63) | h->reset_epoch) : h->t;
  p.seed_lo = (uint32_t)c.seed; p.seed_hi = (uint32_t)(c.seed >> 32);
  p.map_pad = h->map_pad; p.dist_lut = h->dist_lut;
  p.view_table = h->view_table; p.cell_index = h->cell_index;
  p.actions = reinterpret_cast<const uint32_t*>(actions);
  p.order = reinterpret_cast<const uint32_t*>(order);
  p.respawn = reinterpret_cast<const uint32_t*>(respawn);
  p.reset_spawn = reinterpret_cast<const uint32_t*>(reset_spawn);
  p.mask = mask; p.actions_out = reinterpret_cast<uint32_t*>(actions_out);
  p.obs_image = h->buf.obs_image; p.reward = h->buf.reward;
  p.env_reward = h->buf.env_reward; p.dir_done = h->buf.dir_done; p.env_state = h->buf.env_state; p.err = h->err;
  p.kind = c.env_kind; p.GW = h->GW; p.base_grid = h->base_grid;
  p.inv_w20 = ((1 << 20) + c.width - 1) / c.width; p.GB = (h->GW + 3) / 4;
  p.grid0 = h->per_env_maps ? h->grid0 : nullptr;
  p.objdrop = reinterpret_cast<const uint32_t*>(objdrop);
  p.reset_gen = reinterpret_cast<const uint32_t*>(reset_gen);
  p.carry = h->carry; p.grid = h->grid;
  p.n_goals = c.gen_n_goals; p.n_clutter = c.gen_n_clutter; p.gen_flags = c.gen_flags; p.full_obs = h->full_obs ? 1 : 0;
  p.gx = h->gx; p.gy = h->gy; p.LW = h->LW; p.LH = h->LH;
  p.fixed_mode = h->fixed_mode; p.fixed_pd = h->fixed_pd; p.fixed_pd_hi = h->fixed_pd_hi; p.roll_split = 0;
  p.scatter = reinterpret_cast<const uint16_t*>(scatter); p.aux = h->aux; p.obs_full = h->obs_full;
  p.nt_obs = launch_streams_obs(h);
  p.reward_f32 = c.reward_dtype == MG_REWARD_F32 ? 1 : 0;
  int nw; unsigned grid; size_t lds;
  DeviceGuard g(h->device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  if (h->dynamic) {
    if (dest) {
      if (h->full_obs) p.obs_full = dest->obs_image; else p.obs_image = dest->obs_image;
      p.reward = dest->reward; p.dir_done = dest->dir_done;
    }
    if (n_roll > 0) {
      // Fused rollout on mutable grids (mg_dyn_rollout_kernel): the trajectory is written once, so every image is streamed
      p.n_roll = n_roll;
      p.traj_obs = roll->obs_image; p.traj_reward = roll->reward; p.traj_dd = roll->dir_done;
      p.traj_actions = actions ? nullptr : reinterpret_cast<uint32_t*>(roll->actions);
      p.nt_obs = 0xFF;
      if (const char* e = debug_env("MG_DEBUG_NT")) p.nt_obs = atoi(e);   // tuning experiments only
      // small batches: a chunk's agents dealt out over waves, as in the shared-map rollout (mg_rollout_split_kernel); the same
      // kernel does both, told by p.roll_split.  Taken while every chunk's A waves are resident at once (A = 3: up to 32768
      // envs at two waves per SIMD, 65536 for the fully observed kinds' three); beyond that the plain mapping is the faster one.
      bool split = false, one_trip = false;
      if (c.n_agents > 1 && dyn_geometry(h, &nw, &grid, &lds, 1, c.n_agents, &one_trip)) {
        const char* e = debug_env("MG_DEBUG_ROLL_SPLIT");   // tests / A-B timing
        split = e ? atoi(e) != 0 : one_trip;
      }
      if (!split) dyn_geometry(h, &nw, &grid, &lds, 1);
      p.roll_split = split ? 1u : 0u;
      void* kargs[] = {&p};
      HIP_TRY(hipLaunchKernel(h->dyn_roll_fn[dyn_dec_index(h)], dim3(grid), dim3(nw * kWave), kargs, lds, h->stream));
      return launch_ep_stats(h, op, mask, n_roll, roll);
    }
    // the free-running step — no injected draws, no mask, no env_reward output — runs its own instantiation (mg_dyn.h FREE)
    const void* free_fn = h->dyn_free_fn[dyn_dec_index(h)];
    if (free_fn && op == OP_STEP && !order && !respawn && !reset_spawn && !objdrop && !reset_gen && !scatter && !mask && !h->buf.env_reward) {
      // a small batch: the fused rollout kernel with the agents over waves, as a rollout of ONE step into the step's own
      // destinations — the same free-running step, a third of the views per wave (step_split_plan): DoorKey-8x8 x 3 at 4096
      // envs 16.9 -> 13.7 us, 16384 envs 17.9 -> 14.8
      const void* split_fn = nullptr;
      if ((actions != nullptr) != (random_actions != 0) && step_split_plan(h, &split_fn, &nw, &grid, &lds)) {
        p.n_roll = 1; p.roll_split = 1u;
        p.traj_obs = h->full_obs ? p.obs_full : p.obs_image; p.traj_reward = p.reward; p.traj_dd = p.dir_done;
        p.traj_actions = p.actions_out;
        p.obs_full = nullptr; p.obs_image = nullptr; p.reward = nullptr; p.dir_done = nullptr;   // (a rollout's second set of destinations: none)
        void* kargs[] = {&p};
        HIP_TRY(hipLaunchKernel(split_fn, dim3(grid), dim3(nw * kWave), kargs, lds, h->stream));
        return dest ? launch_ep_stats(h, op, mask, 1, dest) : launch_ep_stats(h, op, mask, 0, nullptr);
      }
      dyn_geometry(h, &nw, &grid, &lds, 2);
      void* kargs[] = {&p};
      HIP_TRY(hipLaunchKernel(free_fn, dim3(grid), dim3(nw * kWave), kargs, lds, h->stream));
      return dest ? launch_ep_stats(h, op, mask, 1, dest) : launch_ep_stats(h, op, mask, 0, nullptr);
    }
    dyn_geometry(h, &nw, &grid, &lds);
    hipLaunchKernelGGL(h->dvar->fn[dyn_dec_index(h)], dim3(grid), dim3(nw * kWave), lds, h->stream, p);
    HIP_TRY(hipGetLastError());
    if (op == OP_RESET) h->reset_epoch += 1;
    return dest ? launch_ep_stats(h, op, mask, 1, dest) : launch_ep_stats(h, op, mask, 0, nullptr);
  }
  if (objdrop || reset_gen || scatter) return fail(MG_EINVAL, "objdrop / gen / scatter injection needs a mutable-grid handle");
  if (n_roll > 0) {
    // Fused rollout: the trajectory is written once and far exceeds the caches, so every image is streamed.  Which path:
    // the table kernel (persistent, 16 waves per workgroup, its 44 KB table fill now paid once per K steps) whenever the
    // handle has a table, unless MG_DEBUG_ROLL_PATH forces one (tuning experiments only).
    // (small batches, AUTO: from four steps per launch on — below that the fill costs more than the table saves)
    bool rt = h->use_table && h->var->roll[1] != nullptr && (c.kernel_path == MG_PATH_TABLE || h->n_chunks >= 1024 || n_roll >= 4);
    if (const char* e = debug_env("MG_DEBUG_ROLL_PATH")) rt = rt && atoi(e) == MG_PATH_TABLE;
    p.n_roll = n_roll;
    p.traj_obs = roll->obs_image; p.traj_reward = roll->reward; p.traj_dd = roll->dir_done;
    p.traj_actions = reinterpret_cast<uint32_t*>(roll->actions);
    p.nt_obs = 0xFF; p.random_actions = 1;
    if (const char* e = debug_env("MG_DEBUG_NT")) p.nt_obs = atoi(e);   // tuning experiments only
    launch_geometry(h, rt, &nw, &grid, &lds);
    const size_t fixed = lds_fixed_bytes(h, rt), per_wave = (size_t)kWave * c.view_size * c.view_size * 3;
    const int64_t per_cu_chunks = (h->n_chunks + h->n_cu - 1) / h->n_cu;
    if (!rt) {                                   // persistent on the generic path too: up to four chunks per wave, stepped together
      const int per_cu = std::max(1, std::min((int)((size_t)kLdsLimit / lds), 32 / nw));
      grid = (unsigned)std::min<int64_t>(grid, (int64_t)h->n_cu * per_cu);
    } else if (per_cu_chunks < 16 && !debug_env("MG_DEBUG_NW") && !debug_env("MG_DEBUG_GRID")) {
      // fewer chunks than 16 per CU: one workgroup per CU with just as many waves as the CU's share (any number, not a power of
      // two), so that no wave takes a second chunk while wave slots stand empty
      int w = (int)per_cu_chunks;
      while (w > 1 && fixed + w * per_wave > (size_t)kLdsLimit) --w;
      nw = w; lds = fixed + nw * per_wave;
      const int per_cu = std::max(1, std::min((int)((size_t)kLdsLimit / lds), 32 / nw));
      grid = (unsigned)std::min<int64_t>((h->n_chunks + nw - 1) / nw, (int64_t)h->n_cu * per_cu);
    }
    const bool std_cfg = p.shaping == MG_SHAPING_V8 && p.auto_reset && !p.competitive && h->fixed_mode == MG_FIXED_OFF &&
                         !p.reward_f32 && !c.see_through_walls && !debug_env("MG_DEBUG_ROLL_GENERAL");
    // Small batches: the agents of a chunk dealt out over two to A waves (mg_rollout_split_kernel) — a step is then the latency of
    // one wave, and a wave does a third (half, ...) of the views.  A chunk's waves sit in one workgroup (they meet at a barrier
    // between reading its state and writing it back).  Table path: one workgroup per CU holding the CU's share of chunks, while
    // that share times the waves per chunk fits the workgroup's 16 waves.  Generic path (four waves' staging per workgroup): a
    // chunk per workgroup, while all of them are resident at two waves per SIMD.  Beyond that the action phase done several
    // times over costs throughput, and the plain kernel is the faster one.
    const void* split_fn = h->roll_split_fn[rt ? 1 : 0][std_cfg ? 1 : 0];
    const int A = c.n_agents, cap = rt ? 16 : 4;                          // waves per workgroup
    // waves per chunk: A, while the CU's share of chunks times A fits one workgroup (three agents: up to 5 chunks per CU = 81920
    // envs).  The kernel takes any number from two up (wave `me` does agents me, me + parts, ...); two waves per chunk for 6 to 8
    // chunks per CU were measured and lose to the plain kernel (98304 envs 6.26 against 5.90 us per step, 131072 8.39 against
    // 6.87): from about six chunks per CU the chip is full and the action phase done twice is paid in throughput.
    int parts = 0, cpb = 1;                                               // chunks per workgroup
    if (rt) {
      if (per_cu_chunks * A <= cap && fixed + (size_t)per_cu_chunks * A * per_wave <= (size_t)kLdsLimit) { parts = A; cpb = (int)per_cu_chunks; }
    } else if (A <= cap + 1 && h->n_chunks * A <= (int64_t)h->n_cu * 8) parts = A;
    bool split = split_fn && A > 1 && parts > 0;
    if (const char* e = debug_env("MG_DEBUG_ROLL_SPLIT")) {   // tests / A-B timing: 0 off, 1 on, 2.. on with that many waves per chunk
      split = split_fn && A > 1 && atoi(e) != 0;
      if (split && (atoi(e) > 1 || !parts)) {
        parts = std::min(A, std::max(2, atoi(e) > 1 ? atoi(e) : A));
        cpb = (int)std::max<int64_t>(1, std::min<int64_t>(per_cu_chunks, cap / parts));
        while (cpb > 1 && fixed + (size_t)cpb * parts * per_wave > (size_t)kLdsLimit) --cpb;
      }
    }
    if (split) {
      if (const char* e = debug_env("MG_DEBUG_SPLIT_CPB")) cpb = std::max(1, std::min(atoi(e), std::max(1, cap / parts)));   // tuning experiments only
      p.roll_split = (uint32_t)parts;
      const int snw = cpb * parts;
      const size_t slds = fixed + snw * per_wave;
      const int per_cu = std::max(1, std::min((int)((size_t)kLdsLimit / slds), 32 / snw));
      unsigned sgrid = (unsigned)std::min<int64_t>((h->n_chunks + cpb - 1) / cpb, (int64_t)h->n_cu * per_cu);
      if (const char* e = debug_env("MG_DEBUG_GRID")) sgrid = (unsigned)std::max(1, std::min<int>(sgrid, atoi(e)));   // tests only
      void* kargs[] = {&p};
      HIP_TRY(hipLaunchKernel(split_fn, dim3(sgrid), dim3(snw * kWave), kargs, slds, h->stream));
      return launch_ep_stats(h, op, mask, n_roll, roll);
    }
    hipLaunchKernelGGL((std_cfg ? h->var->roll_std : h->var->roll)[rt ? 1 : 0], dim3(grid), dim3(nw * kWave), lds, h->stream, p);
    HIP_TRY(hipGetLastError());
    return launch_ep_stats(h, op, mask, n_roll, roll);
  }
  const bool table = launch_uses_table(h);
  // A plain step of a small batch: the agents-over-waves kernel as a rollout of ONE step into the handle's own buffers — the
  // same free-running step (mg_rollout_split_kernel is tested against K step launches bit for bit), a third of the views per
  // wave (step_split_plan): 4096 envs 9.80 -> 7.11 us per step launch.
  const void* split_fn = nullptr;
  if (op == OP_STEP && (actions != nullptr) != (random_actions != 0) && !order && !respawn && !reset_spawn && !mask &&
      step_split_plan(h, &split_fn, &nw, &grid, &lds)) {
    p.n_roll = 1; p.roll_split = (uint32_t)c.n_agents;
    p.traj_obs = p.obs_image; p.traj_reward = p.reward; p.traj_dd = p.dir_done; p.traj_actions = p.actions_out;
    p.obs_image = nullptr; p.reward = nullptr; p.dir_done = nullptr;      // (a rollout's second set of destinations: none)
    void* kargs[] = {&p};
    HIP_TRY(hipLaunchKernel(split_fn, dim3(grid), dim3(nw * kWave), kargs, lds, h->stream));
    return launch_ep_stats(h, op, mask, 0, nullptr);
  }
  launch_geometry(h, table, &nw, &grid, &lds);
  const int fast = (op == OP_STEP && !order && !respawn && !reset_spawn && !random_actions && !h->buf.env_reward && !p.reward_f32) ? 1 : 0;
  hipLaunchKernelGGL(table ? h->var->table[fast] : h->var->generic[fast], dim3(grid), dim3(nw * kWave), lds, h->stream, p);
  HIP_TRY(hipGetLastError());
  if (op == OP_RESET) h->reset_epoch += 1;
  return launch_ep_stats(h, op, mask, 0, nullptr);
}

// Switch the handle to the mutable-grid kernel: allocate the per-env grids and carry words (once),
// upload the base layout every reset starts from and the shaper's distance LUT for goal (gx, gy).
int enable_dynamic(mg_handle* h, const std::vector<uint8_t>& kinds, int gx, int gy) {
  const mg_config& c = h->cfg;
  if (!h->dvar) return fail(MG_EUNSUPPORTED, "no mutable-grid kernel instantiated for this (n_agents, view_size)");
  DeviceGuard g(h->device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  {
    const int lw0 = h->LW, lh0 = h->LH;
    h->LW = std::max(c.width, gx + 1); h->LH = std::max(c.height, gy + 1);
    int nw; unsigned grid; size_t lds;
    dyn_geometry(h, &nw, &grid, &lds);
    h->LW = lw0; h->LH = lh0;
    if (lds > (size_t)kLdsLimit) return fail(MG_EUNSUPPORTED, "grid too large for the mutable-grid kernel's LDS budget");
  }
  int rc = MG_OK;
  if (!h->grid) {
    if (rc == MG_OK) rc = dalloc(h, &h->base_grid, (size_t)h->GW);
    if (rc == MG_OK) rc = dalloc(h, &h->carry, (size_t)h->stride * (h->cfg.n_agents > 4 ? 2 : 1));
    if (rc == MG_OK) rc = dalloc(h, &h->grid, (size_t)h->n_chunks * ((h->GW + 3) / 4 * 4) * kWave);
    if (rc != MG_OK) return rc;
  }
  std::vector<uint32_t> base((size_t)h->GW, 0u);       // the layout as cell codes (mg_dyn.h), from public kind ids
  for (size_t i = 0; i < kinds.size(); ++i) base[i >> 2] |= code_from_kind(kinds[i]) << (8 * (i & 3));
  // relative distance LUT: d[|dy|*LW + |dx|] = np.linalg.norm((dx, dy)) (v8:192).  (gx, gy) can lie outside a
  // small grid (the (13,13) fallback of get_goal_position, v8:167), so the table covers max(W, gx+1) x max(H, gy+1).
  const int LW = std::max(c.width, gx + 1), LH = std::max(c.height, gy + 1);
  if ((size_t)LW * LH > (size_t)64 * 64) return fail(MG_EINVAL, "goal too far outside the grid");
  std::vector<double> d((size_t)LW * LH);
  for (int dy = 0; dy < LH; ++dy)
    for (int dx = 0; dx < LW; ++dx) d[(size_t)dy * LW + dx] = sqrt((double)((long)dx * dx + (long)dy * dy));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->base_grid, base.data(), base.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dist_lut, d.data(), d.size() * 8, hipMemcpyHostToDevice));
  h->gx = gx; h->gy = gy; h->LW = LW; h->LH = LH;
  h->dynamic = true; h->use_table = false; h->n_entries = 0;
  h->have_map = true;
  h->have_reset = false;          // positions and grids of the old layout mean nothing on the new one
  h->buf.carry = h->carry; h->buf.grid = h->grid; h->buf.grid_words = (h->GW + 3) / 4 * 4; h->buf.mutable_grid = 1;
  return MG_OK;
}

}  // namespace

namespace { int noalloc_warm(int device); }

extern "C" {

const char* mg_last_error(void) { return g_err.c_str(); }
const char* mg_version(void) { return "mgrid-hip 0.3 (gfx950)"; }

int mg_create(const mg_config* cfg, int device, mg_handle** out) {
  if (!cfg || !out) return fail(MG_EINVAL, "null argument");
  *out = nullptr;
  const mg_config& c = *cfg;
  if (c.width < 3 || c.height < 3 || c.width > 64 || c.height > 64) return fail(MG_EINVAL, "width/height must be in [3,64]");
  if (c.n_envs < 1) return fail(MG_EINVAL, "n_envs must be >= 1");
  if (c.max_steps < 1 || c.max_steps > 65534) return fail(MG_EINVAL, "max_steps must be in [1,65534]");
  if (c.shaping_kind < MG_SHAPING_NONE || c.shaping_kind > MG_SHAPING_V7) return fail(MG_EINVAL, "unknown shaping_kind");
  if (c.kernel_path < MG_PATH_AUTO || c.kernel_path > MG_PATH_TABLE) return fail(MG_EINVAL, "unknown kernel_path");
  if (c.reward_dtype != MG_REWARD_F64 && c.reward_dtype != MG_REWARD_F32) return fail(MG_EINVAL, "unknown reward_dtype");
  if (c.reward_dtype == MG_REWARD_F32 && c.emit_env_reward) return fail(MG_EUNSUPPORTED, "env_reward is a float64 output: reward_dtype must be MG_REWARD_F64 with emit_env_reward");
  if (c.env_id_base < 0 || (uint64_t)c.env_id_base + (uint64_t)c.n_envs > 0xFFFFFFFFull)
    return fail(MG_EINVAL, "global env ids must fit 32 bits");
  if (c.env_kind < MG_KIND_STATIC || c.env_kind > MG_KIND_FOURROOMS) return fail(MG_EINVAL, "unknown env_kind");
  const Variant* var = nullptr;
  for (const Variant& v : kVariants)
    if (v.A == c.n_agents && v.V == c.view_size) var = &v;
  // fully observed kinds never compute a view: they have instantiations of their own, whatever agent_view_size says
  const bool want_full = c.env_kind == MG_KIND_MEETUP;
  const DynVariant* dvar = nullptr;
  // among the matching instantiations: the smallest prefetch depth that covers the grid, else the deepest there is
  const int gw = (c.width * c.height + 3) / 4;
  for (const DynVariant& v : kDynVariants) {
    if (v.A != c.n_agents || v.full != want_full || !(want_full || v.V == c.view_size)) continue;
    if (!dvar || (dvar->pre < gw && v.pre > dvar->pre) || (v.pre >= gw && v.pre < dvar->pre)) dvar = &v;
  }
  if (c.env_kind == MG_KIND_STATIC) {
    if (!var) return fail(MG_EUNSUPPORTED, "no kernel instantiated for this (n_agents, view_size)");
    if (c.kernel_path == MG_PATH_TABLE && !var->table[0]) return fail(MG_EUNSUPPORTED, "no table kernel for this view_size");
  } else {
    if (!dvar) return fail(MG_EUNSUPPORTED, "no mutable-grid kernel instantiated for this (n_agents, view_size)");
    if (c.kernel_path == MG_PATH_TABLE) return fail(MG_EUNSUPPORTED, "no table kernel for mutable grids");
    if (c.env_kind == MG_KIND_DOORKEY) {
      if (c.width != c.height || c.width < 5) return fail(MG_EINVAL, "DoorKey needs a square grid of size >= 5");
    } else if (c.env_kind == MG_KIND_FOURROOMS) {
      if (c.width < 6 || c.height < 6) return fail(MG_EINVAL, "FourRooms needs a grid of at least 6x6");
    } else {
      const int interior = (c.width - 2) * (c.height - 2);
      if (c.gen_n_clutter < 0 || c.gen_n_goals < 0 || c.gen_n_goals + c.gen_n_clutter > MG_MAX_SCATTER)
        return fail(MG_EINVAL, "gen_n_goals + gen_n_clutter must be in [0, 64]");
      if (c.env_kind == MG_KIND_CLUTTERED && c.gen_n_goals > 1) return fail(MG_EINVAL, "Cluttered has one goal: gen_n_goals is 0 or 1");
      if (c.env_kind == MG_KIND_MEETUP && c.n_agents > 4) return fail(MG_EUNSUPPORTED, "Meetup supports up to four agents");
      if (c.env_kind == MG_KIND_MEETUP && (c.gen_n_goals < 1 || c.gen_n_goals > kMaxGoals))
        return fail(MG_EUNSUPPORTED, "Meetup supports 1 to 6 goals");
      if (c.gen_n_goals + c.gen_n_clutter + 1 + c.n_agents > interior) return fail(MG_EINVAL, "more objects than interior cells");
    }
    var = nullptr;
  }

  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MG_EINVAL, "bad device index");
  DeviceGuard g(device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");

  (void)noalloc_warm(device);   // mg_step_noalloc's per-device set-up, so that its own calls find it done (a failure is reported there)
  mg_handle* h = new mg_handle();
  h->cfg = c; h->device = device; h->var = var; h->t = 0; h->reset_epoch = 0; h->have_map = h->have_reset = false;
  h->dynamic = false; h->dvar = dvar; h->GW = (c.width * c.height + 3) / 4;
  h->base_grid = nullptr; h->grid0 = nullptr; h->per_env_maps = false; h->carry = nullptr; h->grid = nullptr; h->aux = nullptr; h->obs_full = nullptr;
  h->full_obs = c.env_kind == MG_KIND_MEETUP; h->LW = c.width; h->LH = c.height;
  h->ep_on = false; h->ep_thr = 0.0; memset(&h->ep, 0, sizeof(h->ep));
  h->dyn_waves_cu2[0] = h->dyn_waves_cu2[1] = 8; h->dyn_dec = false; h->fixed_mode = 0; h->fixed_pd = 0; h->fixed_pd_hi = 0;
  h->dyn_roll_waves_cu2[0] = h->dyn_roll_waves_cu2[1] = 8;
  h->dyn_free_waves_cu2[0] = h->dyn_free_waves_cu2[1] = 8;
  for (int t = 0; t < 2; ++t)
    for (int sc = 0; sc < 2; ++sc) h->roll_split_fn[t][sc] = var ? mg_rollsplit_kernel(var->A, var->V, t, sc) : nullptr;
  for (int v = 0; v < 2; ++v) {
    h->dyn_roll_fn[v] = dvar ? mg_dynroll_kernel(dvar->A, dvar->V, dvar->full ? 1 : 0, dvar->pre, v, 0) : nullptr;
    h->dyn_free_fn[v] = dvar ? mg_dynroll_kernel(dvar->A, dvar->V, dvar->full ? 1 : 0, dvar->pre, v, 1) : nullptr;
  }
  if (debug_env("MG_DEBUG_DYN_NO_FREE")) h->dyn_free_fn[0] = h->dyn_free_fn[1] = nullptr;     // tests / A-B timing: the general step kernel
  for (int v = 0; dvar && v < 6; ++v) {
    hipFuncAttributes fa;
    const void* fn = v < 2 ? reinterpret_cast<const void*>(dvar->fn[v]) : v < 4 ? h->dyn_roll_fn[v - 2] : h->dyn_free_fn[v - 4];
    if (fn && hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0) {
      const int regs = (fa.numRegs + 7) / 8 * 8;                  // gfx950: 512 VGPRs per SIMD lane, allocated in 8s
      (v < 2 ? h->dyn_waves_cu2[v] : v < 4 ? h->dyn_roll_waves_cu2[v - 2] : h->dyn_free_waves_cu2[v - 4]) = 4 * std::max(1, std::min(8, 512 / regs));
    }
  }
  // DoorKey's door row rides in five bits of the state word (mg_dyn.h): grids up to 33 rows; Cluttered / FourRooms
  // layouts never hold a door; a static layout decides in mg_set_static_map
  h->dyn_dec = (c.env_kind == MG_KIND_DOORKEY && c.height <= 33) || c.env_kind == MG_KIND_CLUTTERED || c.env_kind == MG_KIND_FOURROOMS;
  h->use_table = false; h->n_entries = 0; h->view_table = nullptr; h->cell_index = nullptr;
  h->stride = (c.n_envs + 63) / 64 * 64;
  h->n_chunks = h->stride / 64;
  h->dd_bytes = (c.n_agents + 1 <= 4) ? 4 : 8;
  h->PW = c.width + 2 * (c.view_size - 1);
  h->PH = c.height + 2 * (c.view_size - 1);
  h->gx = h->gy = 13;
  h->n_cu = 256;
  (void)hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device);
  if (h->n_cu < 1) h->n_cu = 1;
  const int rowb = c.view_size * c.view_size * 3;
  h->own_stream = true;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return fail(MG_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  for (kernel_fn fn : {var ? var->generic[0] : nullptr, var ? var->generic[1] : nullptr, var ? var->table[0] : nullptr,
                       var ? var->table[1] : nullptr, var ? var->roll[0] : nullptr, var ? var->roll[1] : nullptr,
                       var ? var->roll_std[0] : nullptr, var ? var->roll_std[1] : nullptr,
                       dvar ? dvar->fn[0] : nullptr, dvar ? dvar->fn[1] : nullptr}) {
    if (!fn) continue;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    if (e != hipSuccess) { (void)hipStreamDestroy(h->stream); delete h; return fail(MG_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }
  }

  const size_t A = (size_t)c.n_agents, S = (size_t)h->stride;
  memset(&h->buf, 0, sizeof(h->buf));
  int rc = MG_OK;
  if (rc == MG_OK) rc = dalloc(h, &h->map_pad, (size_t)h->PW * h->PH);
  if (rc == MG_OK) rc = dalloc(h, &h->dist_lut, (size_t)64 * 64);
  if (rc == MG_OK) rc = dalloc(h, &h->cell_index, (size_t)c.width * c.height);
  if (rc == MG_OK && var) rc = dalloc(h, &h->view_table, (size_t)c.width * c.height * 4 * (var->entw ? var->entw : 4));
  if (rc == MG_OK) rc = dalloc(h, &h->err, 1);
  if (rc == MG_OK && !h->full_obs) rc = dalloc(h, &h->buf.obs_image, A * S * rowb);
  if (rc == MG_OK && h->full_obs) rc = dalloc(h, &h->obs_full, S * (size_t)c.width * c.height * 3);
  if (rc == MG_OK && (c.env_kind == MG_KIND_MEETUP || c.env_kind == MG_KIND_CLUTTERED || c.env_kind == MG_KIND_FOURROOMS))
    rc = dalloc(h, &h->aux, S * 4);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.reward, A * S);
  if (rc == MG_OK && c.emit_env_reward) rc = dalloc(h, &h->buf.env_reward, A * S);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.dir_done, S * (size_t)h->dd_bytes);
  if (rc == MG_OK) rc = dalloc(h, &h->buf.env_state, (A + 1) * S);
  if (rc != MG_OK) {
    for (void* q : h->allocs) (void)hipFree(q);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return rc;
  }
  h->buf.n_envs = c.n_envs; h->buf.env_stride = h->stride; h->buf.n_agents = c.n_agents; h->buf.obs_cell_bytes = rowb;
  h->buf.dir_done_bytes = h->dd_bytes;
  h->buf.reward_elem_bytes = c.reward_dtype == MG_REWARD_F32 ? 4 : 8;
  h->buf.aux = h->aux; h->buf.obs_full = h->obs_full;
  h->buf.obs_full_bytes = h->full_obs ? c.width * c.height * 3 : 0;
  if (c.env_kind != MG_KIND_STATIC) {
    // the part of _gen_grid that is the same for every env and every reset: border walls (wall_rect) and,
    // for DoorKey (doorkey.py:38-44) and Cluttered without randomize_goal (cluttered.py:31), the goal at
    // (W-2, H-2); the rest is drawn on device
    std::vector<uint8_t> kinds((size_t)c.width * c.height, 0);
    for (int y = 0; y < c.height; ++y)
      for (int x = 0; x < c.width; ++x)
        if (x == 0 || y == 0 || x == c.width - 1 || y == c.height - 1) kinds[(size_t)y * c.width + x] = (uint8_t)MG_KIND_ID(MG_T_WALL, 5, 0);
    const bool fixed_goal = c.env_kind == MG_KIND_DOORKEY || (c.env_kind == MG_KIND_CLUTTERED && c.gen_n_goals == 0);
    if (fixed_goal) kinds[(size_t)(c.height - 2) * c.width + (c.width - 2)] = (uint8_t)MG_KIND_ID(MG_T_GOAL, 1, 0);
    // shaper goal (v8:159-169): that cell; Meetup has no goal cell -> the (13,13) fallback; a scattered Cluttered
    // goal is per env (mg_buffers.aux word 0) and overrides this in the kernel
    const bool no_goal = c.env_kind == MG_KIND_MEETUP;
    rc = enable_dynamic(h, kinds, no_goal ? 13 : c.width - 2, no_goal ? 13 : c.height - 2);
    if (rc != MG_OK) { mg_destroy(h); return rc; }
  }
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
  const mg_config& cfg = h->cfg;
  if (cfg.env_kind != MG_KIND_STATIC) return fail(MG_EINVAL, "this env_kind generates its own grids; no static map");
  const int W = cfg.width, H = cfg.height, V = cfg.view_size, pad = V - 1;
  {
    // any object an agent can pick up, toggle or walk onto and destroy -> per-env mutable grids
    bool objects = false, doors = false;
    std::vector<uint8_t> kinds((size_t)W * H, 0);
    h->free_cells.assign((size_t)W * H, 0);
    for (int i = 0; i < W * H; ++i) h->free_cells[i] = cells[(size_t)i * 3] <= MG_T_EMPTY;
    h->fixed_mode = 0;                      // start cells belong to the old layout
    h->per_env_maps = false;                // and so do per-env layouts (mg_set_maps)
    int n_free = 0, gx = -1, gy = -1;
    for (int i = 0; i < W * H; ++i) {
      const uint8_t* c = cells + (size_t)i * 3;
      switch (c[0]) {
        case MG_T_UNSEEN: case MG_T_EMPTY: ++n_free; break;
        case MG_T_WALL: case MG_T_GOAL: case MG_T_LAVA: kinds[i] = (uint8_t)(c[0] * 8 + (c[1] & 7)); break;
        case MG_T_FLOOR: case MG_T_KEY: case MG_T_BALL: case MG_T_BOX:
          kinds[i] = (uint8_t)(c[0] * 8 + (c[1] & 7)); objects = true; break;
        case MG_T_DOOR:
          if (c[2] > 2) return fail(MG_EINVAL, "door state must be 0 (open), 1 (closed) or 2 (locked)");
          kinds[i] = (uint8_t)(128 + (c[1] & 7) * 4 + c[2]); objects = doors = true; break;
        default: return fail(MG_EUNSUPPORTED, "unknown object type in the map");
      }
    }
    if (objects) {
      if (n_free < cfg.n_agents) return fail(MG_EINVAL, "fewer free cells than agents");
      for (int i = 0; i < W && gx < 0; ++i)      // get_goal_position (v8_robust_ppo.py:159-169)
        for (int j = 0; j < H; ++j)
          if (cells[((size_t)j * W + i) * 3] == MG_T_GOAL) { gx = i; gy = j; break; }
      if (gx < 0) { gx = 13; gy = 13; }
      h->dyn_dec = !doors;     // a door's closed / locked state is not in its cell code: such layouts decode through the table
      return enable_dynamic(h, kinds, gx, gy);
    }
    h->dynamic = false;
    h->buf.carry = nullptr; h->buf.grid = nullptr; h->buf.grid_words = 0; h->buf.mutable_grid = 0;
  }
  const uint32_t wall = (uint32_t)MG_T_WALL | (5u << 8) | F_OPAQUE;   // Grid.slice's out-of-bounds Wall() (multigrid.py:315)
  std::vector<uint32_t> m((size_t)h->PW * h->PH, wall);
  std::vector<uint16_t> cidx((size_t)W * H, 0);
  int gx = -1, gy = -1, n_free = 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const uint8_t* c = cells + ((size_t)y * W + x) * 3;
      uint32_t w = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
      switch (c[0]) {
        case MG_T_UNSEEN: case MG_T_EMPTY: w = (uint32_t)MG_T_EMPTY | F_FREE; cidx[(size_t)y * W + x] = (uint16_t)n_free++; break;
        case MG_T_WALL: w |= F_OPAQUE; break;
        case MG_T_GOAL: w |= F_KILL | F_GOAL; break;
        case MG_T_LAVA:          // grey lava = LavaWall (lava_walls.py:25-38): it kills, it is transparent, it looks like a grey wall
          if (c[1] == 5) w = (uint32_t)MG_T_WALL | (5u << 8);
          w |= F_KILL;
          break;
        default:
          return fail(MG_EUNSUPPORTED, "static maps hold only empty/wall/goal/lava cells");
      }
      m[(size_t)(y + pad) * h->PW + (x + pad)] = w;
    }
  if (n_free < cfg.n_agents) return fail(MG_EINVAL, "fewer free cells than agents");
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
  // Static view table: the view from every free cell in every direction (agents are transparent
  // and nothing else moves on a static map, so slice+rotate+process_vis+encode of the static
  // part is a pure function of (cell, dir)).
  const Variant* var = h->var;
  std::vector<uint32_t> table;
  // AUTO: the handle has a table whatever its size — a fused rollout pays the fill once per K steps and takes the table path
  // at any batch size; which kernel a step launch runs is launch_uses_table()'s decision
  bool use_table = var->table[0] != nullptr && cfg.kernel_path != MG_PATH_GENERIC;
  if (use_table) {
    h->n_entries = n_free * 4;
    table.assign((size_t)h->n_entries * var->entw, 0u);
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        if (!(m[(size_t)(y + pad) * h->PW + (x + pad)] & F_FREE)) continue;
        for (int dir = 0; dir < 4; ++dir) {
          uint32_t* out = table.data() + ((size_t)cidx[(size_t)y * W + x] * 4 + dir) * var->entw;
          if (V == 5) host_view_entry<5>(m, h->PW, pad, cfg.see_through_walls != 0, x, y, dir, out);
          else host_view_entry<3>(m, h->PW, pad, cfg.see_through_walls != 0, x, y, dir, out);
        }
      }
    // the table, the other LUTs and at least one wave's staging must fit the CU's LDS
    const size_t need = lds_fixed_bytes(h, true) + (size_t)kWave * V * V * 3;
    if (need > (size_t)kLdsLimit) {
      if (cfg.kernel_path == MG_PATH_TABLE) return fail(MG_EUNSUPPORTED, "view table does not fit LDS for this map");
      use_table = false;
    }
  }
  if (!use_table) h->n_entries = 0;
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->map_pad, m.data(), m.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dist_lut, d.data(), d.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->cell_index, cidx.data(), cidx.size() * 2, hipMemcpyHostToDevice));
  if (use_table) HIP_TRY(hipMemcpy(h->view_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
  h->use_table = use_table;
  h->have_map = true;
  h->have_reset = false;          // positions and grids of the old layout mean nothing on the new one
  return MG_OK;
}

int mg_set_maps(mg_handle* h, const uint8_t* cells) {
  if (!h || !cells) return fail(MG_EINVAL, "null argument");
  const mg_config& c = h->cfg;
  if (c.env_kind != MG_KIND_STATIC) return fail(MG_EINVAL, "this env_kind generates its own grids; no caller-supplied maps");
  // every env gets its own (mutable) grid; the reset template is per env, the shared template is the bare border
  std::vector<uint8_t> kinds((size_t)c.width * c.height, 0);
  h->dyn_dec = false;                             // the layouts are on the device and may hold doors: table decode
  int rc = enable_dynamic(h, kinds, 13, 13);      // distance table large enough for the (13,13) fallback goal
  if (rc != MG_OK) return rc;
  DeviceGuard g(h->device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  if (!h->grid0) rc = dalloc(h, &h->grid0, (size_t)h->n_chunks * ((h->GW + 3) / 4 * 4) * kWave);
  if (rc == MG_OK && !h->aux) rc = dalloc(h, &h->aux, (size_t)h->stride * 4);
  if (rc != MG_OK) return rc;
  h->buf.aux = h->aux;
  const unsigned blocks = (unsigned)((c.n_envs + 255) / 256);
  hipLaunchKernelGGL(import_maps_kernel, dim3(blocks), dim3(256), 0, h->stream, cells, c.width, c.height, (h->GW + 3) / 4, c.n_agents, c.n_envs,
                     reinterpret_cast<uint8_t*>(h->grid0), h->aux, h->err);
  HIP_TRY(hipGetLastError());
  h->free_cells.assign((size_t)c.width * c.height, 0);     // no fixed start cells on per-env layouts
  h->fixed_mode = 0;
  h->per_env_maps = true;
  return MG_OK;
}

int mg_set_fixed_spawn(mg_handle* h, const int8_t* xy, int mode) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (mode == MG_FIXED_OFF) { h->fixed_mode = 0; return MG_OK; }
  if (!xy || (mode != MG_FIXED_ALWAYS && mode != MG_FIXED_AT_RESET)) return fail(MG_EINVAL, "bad argument");
  if (h->cfg.env_kind != MG_KIND_STATIC || !h->have_map) return fail(MG_ESTATE, "fixed start cells need a static map (mg_set_static_map first)");
  if (h->dynamic) return fail(MG_EUNSUPPORTED, "fixed start cells are not implemented for mutable grids");
  const int W = h->cfg.width, H = h->cfg.height, A = h->cfg.n_agents;
  uint64_t pd = 0;
  uint32_t pd_hi = 0;
  for (int a = 0; a < A; ++a) {
    const int x = xy[2 * a], y = xy[2 * a + 1];
    if (x < 0 || y < 0 || x >= W || y >= H || !h->free_cells[(size_t)y * W + x]) return fail(MG_EINVAL, "start cell is not a free cell of the map");
    for (int b = 0; b < a; ++b)
      if (xy[2 * b] == x && xy[2 * b + 1] == y) return fail(MG_EINVAL, "two agents share a start cell");
    const uint32_t w = (uint32_t)x | ((uint32_t)y << 6);
    if (a < 4) pd |= (uint64_t)w << (14 * a); else pd_hi = w;
  }
  h->fixed_pd = pd; h->fixed_pd_hi = pd_hi; h->fixed_mode = mode; h->have_reset = false;
  return MG_OK;
}

int mg_reset(mg_handle* h, const uint8_t* env_mask) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (!h->have_map) return fail(MG_ESTATE, "mg_set_static_map must be called before mg_reset");
  int rc = launch(h, OP_RESET, nullptr, nullptr, nullptr, nullptr, env_mask, 0, nullptr);
  if (rc == MG_OK) h->have_reset = true;
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

int mg_reset_inject_dyn(mg_handle* h, const uint8_t* env_mask, const int8_t* spawn, const int8_t* gen) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (!h->have_map) return fail(MG_ESTATE, "mg_set_static_map must be called before mg_reset_inject_dyn");
  if (!h->dynamic) return fail(MG_EINVAL, "not a mutable-grid handle");
  int rc = launch(h, OP_RESET, nullptr, nullptr, nullptr, spawn, env_mask, 0, nullptr, nullptr, gen);
  if (rc == MG_OK) h->have_reset = true;
  return rc;
}

int mg_step_replay_dyn(mg_handle* h, const void* actions, const uint8_t* order, const int8_t* respawn,
                       const int8_t* objdrop, const int8_t* reset_spawn, const int8_t* reset_gen) {
  if (!h || !actions) return fail(MG_EINVAL, "null argument");
  if (!h->dynamic) return fail(MG_EINVAL, "not a mutable-grid handle");
  if (!h->have_reset) return fail(MG_ESTATE, "mg_reset must be called before mg_step_replay_dyn");
  int rc = launch(h, OP_STEP, actions, order, respawn, reset_spawn, nullptr, 0, nullptr, objdrop, reset_gen);
  if (rc == MG_OK) h->t += 1;
  return rc;
}

int mg_reset_inject_scatter(mg_handle* h, const uint8_t* env_mask, const int8_t* spawn, const int8_t* scatter) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (h->cfg.env_kind != MG_KIND_CLUTTERED && h->cfg.env_kind != MG_KIND_MEETUP && h->cfg.env_kind != MG_KIND_FOURROOMS)
    return fail(MG_EINVAL, "scatter injection is for the CLUTTERED / MEETUP / FOURROOMS kinds");
  int rc = launch(h, OP_RESET, nullptr, nullptr, nullptr, spawn, env_mask, 0, nullptr, nullptr, nullptr, scatter);
  if (rc == MG_OK) h->have_reset = true;
  return rc;
}

int mg_export_grid(mg_handle* h, uint8_t* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  if (!h->dynamic) return fail(MG_EINVAL, "not a mutable-grid handle");
  DeviceGuard g(h->device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  const int cells = h->cfg.width * h->cfg.height;
  const int64_t total = h->cfg.n_envs * cells;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(export_grid_kernel, dim3(blocks), dim3(256), 0, h->stream, reinterpret_cast<const uint8_t*>(h->grid), (h->GW + 3) / 4,
                     cells, h->cfg.n_envs, out);
  HIP_TRY(hipGetLastError());
  return MG_OK;
}

int mg_step_random(mg_handle* h, uint8_t* actions_out) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (!h->have_reset) return fail(MG_ESTATE, "mg_reset must be called before mg_step_random");
  int rc = launch(h, OP_STEP, nullptr, nullptr, nullptr, nullptr, nullptr, 1, actions_out);
  if (rc == MG_OK) h->t += 1;
  return rc;
}

// ---- mg_step_noalloc ---------------------------------------------------------------------------
}  // extern "C"

namespace {

// Per-device set-up of the allocation-free entry point: the address of THIS device's copy of g_scratch (every GPU holds
// its own copy of a __device__ array at its own address), its CU count, and the LDS attribute of the kernels it may
// launch.  Done once per device, with that device current; afterwards host code uses `ptr` only.
constexpr int kMaxDevices = 64;
struct NoAllocDevice { bool ready; unsigned char* ptr; int n_cu; };
NoAllocDevice g_noalloc[kMaxDevices];
pthread_mutex_t g_noalloc_mu = PTHREAD_MUTEX_INITIALIZER;

int noalloc_device(int device, NoAllocDevice* out) {
  if (device < 0 || device >= kMaxDevices) return fail(MG_EINVAL, "bad device index");
  pthread_mutex_lock(&g_noalloc_mu);
  NoAllocDevice& d = g_noalloc[device];
  int rc = MG_OK;
  if (!d.ready) {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_scratch));   // needs `device` current: the caller's DeviceGuard
    if (e != hipSuccess || !ptr) rc = fail(MG_EHIP, std::string("hipGetSymbolAddress(g_scratch): ") + hipGetErrorString(e));
    int n_cu = 256;
    if (rc == MG_OK) {
      (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
      for (const Variant& v : kVariants) {
        for (kernel_fn fn : {v.generic[0], v.generic[1]}) {
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
          if (e != hipSuccess) rc = fail(MG_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        }
        if (rc != MG_OK) break;
      }
    }
    if (rc == MG_OK) { d.ptr = static_cast<unsigned char*>(ptr); d.n_cu = n_cu < 1 ? 1 : n_cu; d.ready = true; }
  }
  if (rc == MG_OK) *out = d;
  pthread_mutex_unlock(&g_noalloc_mu);
  return rc;
}

int noalloc_warm(int device) { NoAllocDevice d; return noalloc_device(device, &d); }

}  // namespace

extern "C" {

int mg_step_noalloc(const mg_config* cfg, int device, void* hip_stream, uint64_t step_index, const uint8_t* cells,
                    const void* actions, uint32_t* env_state, uint8_t* obs_image, double* reward, uint8_t* dir_done,
                    uint32_t* err_out) {
  if (!cfg || !cells || !actions || !env_state || !obs_image || !reward || !dir_done) return fail(MG_EINVAL, "null argument");
  const mg_config& c = *cfg;
  // sizes beyond what the per-GPU scratch array was laid out for
  if (c.width < 3 || c.height < 3 || c.width > 64 || c.height > 64)
    return fail(MG_EINVAL, "mg_step_noalloc: width/height must be in [3,64] (the scratch array holds one 64x64 distance table)");
  if (c.view_size != 3 && c.view_size != 5 && c.view_size != 7)
    return fail(MG_EUNSUPPORTED, "mg_step_noalloc: view_size must be 3, 5 or 7 (the scratch array holds a map padded by at most 6 cells)");
  if (c.env_kind != MG_KIND_STATIC) return fail(MG_EUNSUPPORTED, "mg_step_noalloc: shared static layouts only (env_kind = MG_KIND_STATIC)");
  if (c.emit_env_reward) return fail(MG_EUNSUPPORTED, "mg_step_noalloc: no env_reward output");
  if (c.n_envs < 1) return fail(MG_EINVAL, "n_envs must be >= 1");
  if (c.max_steps < 1 || c.max_steps > 65534) return fail(MG_EINVAL, "max_steps must be in [1,65534]");
  if (c.shaping_kind < MG_SHAPING_NONE || c.shaping_kind > MG_SHAPING_V7) return fail(MG_EINVAL, "unknown shaping_kind");
  if (c.reward_dtype != MG_REWARD_F64 && c.reward_dtype != MG_REWARD_F32) return fail(MG_EINVAL, "unknown reward_dtype");
  if (c.env_id_base < 0 || (uint64_t)c.env_id_base + (uint64_t)c.n_envs > 0xFFFFFFFFull)
    return fail(MG_EINVAL, "global env ids must fit 32 bits");
  const Variant* var = nullptr;
  for (const Variant& v : kVariants)
    if (v.A == c.n_agents && v.V == c.view_size) var = &v;
  if (!var) return fail(MG_EUNSUPPORTED, "no kernel instantiated for this (n_agents, view_size)");
  const int PW = c.width + 2 * (c.view_size - 1), PH = c.height + 2 * (c.view_size - 1);
  if (kScratchMap + (size_t)PW * PH * 4 > kScratchBytes || (size_t)c.width * c.height * 8 > kScratchMap - kScratchDist)
    return fail(MG_EINVAL, "mg_step_noalloc: layout does not fit the scratch array");

  DeviceGuard g(device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  NoAllocDevice d;
  int rc = noalloc_device(device, &d);
  if (rc != MG_OK) return rc;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  uint32_t* err = reinterpret_cast<uint32_t*>(d.ptr);
  double* dist = reinterpret_cast<double*>(d.ptr + kScratchDist);
  uint32_t* map = reinterpret_cast<uint32_t*>(d.ptr + kScratchMap);

  // the array keeps the last call's contents: the error word starts at zero, map and table are rewritten in full
  HIP_TRY(hipMemsetAsync(err, 0, 4, stream));
  hipLaunchKernelGGL(noalloc_prep, dim3(1), dim3(256), 0, stream, cells, c.width, c.height, c.view_size, c.n_agents, err, dist, map);
  HIP_TRY(hipGetLastError());

  mg_handle hh;                      // geometry helpers read a handle; this one lives on the stack and owns nothing
  hh.cfg = c; hh.device = device; hh.var = var; hh.use_table = false; hh.n_entries = 0;
  hh.stride = (c.n_envs + 63) / 64 * 64; hh.n_chunks = hh.stride / 64; hh.PW = PW; hh.PH = PH; hh.n_cu = d.n_cu;
  hh.dynamic = false; hh.full_obs = false;
  hh.cfg.kernel_path = MG_PATH_GENERIC;
  Params p;
  memset(&p, 0, sizeof(p));
  p.W = c.width; p.H = c.height; p.max_steps = c.max_steps; p.see_through = c.see_through_walls;
  p.competitive = (c.n_agents == 1) ? 1 : c.competitive;
  p.shaping = c.shaping_kind; p.auto_reset = c.auto_reset; p.dd_bytes = (c.n_agents + 1 <= 4) ? 4 : 8;
  p.op = OP_STEP; p.PW = PW; p.PH = PH;
  p.n_envs = c.n_envs; p.stride = hh.stride; p.n_chunks = hh.n_chunks;
  p.gid_base = (uint64_t)c.env_id_base; p.t = step_index;
  p.seed_lo = (uint32_t)c.seed; p.seed_hi = (uint32_t)(c.seed >> 32);
  p.map_pad = map; p.dist_lut = dist;
  p.actions = reinterpret_cast<const uint32_t*>(actions);
  p.obs_image = obs_image; p.reward = reward; p.dir_done = dir_done; p.env_state = env_state; p.err = err;
  p.nt_obs = hh.n_chunks > (int64_t)d.n_cu * 16 * kUpfront ? 0xFF : 0;
  p.reward_f32 = c.reward_dtype == MG_REWARD_F32 ? 1 : 0;
  int nw; unsigned grid; size_t lds;
  launch_geometry(&hh, false, &nw, &grid, &lds);
  hipLaunchKernelGGL(var->generic[p.reward_f32 ? 0 : 1], dim3(grid), dim3(nw * kWave), lds, stream, p);   // FAST is float64-only
  HIP_TRY(hipGetLastError());
  if (err_out) HIP_TRY(hipMemcpyAsync(err_out, err, 4, hipMemcpyDeviceToDevice, stream));
  return MG_OK;
}

// Test hook: sqrt(k) for k = 0..n as noalloc_prep evaluates it on the device, copied to host `out` (n + 1 doubles).
int mg_debug_device_sqrt(int device, int32_t n, double* out) {
  if (!out || n < 0) return fail(MG_EINVAL, "bad argument");
  DeviceGuard g(device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  double* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, ((size_t)n + 1) * 8));
  hipLaunchKernelGGL(sqrt_probe, dim3(64), dim3(256), 0, 0, (int)n, d);
  hipError_t e = hipMemcpy(out, d, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MG_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  return MG_OK;
}

int mg_rollout_random(mg_handle* h, int32_t n_steps, const mg_rollout_out* out) { return mg_rollout(h, n_steps, nullptr, out); }

int mg_rollout(mg_handle* h, int32_t n_steps, const void* actions, const mg_rollout_out* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  if (n_steps < 1 || n_steps > 65535) return fail(MG_EINVAL, "n_steps must be in [1,65535]");
  if (!out->obs_image || !out->reward || !out->dir_done) return fail(MG_EINVAL, "obs_image, reward and dir_done of mg_rollout_out must be set");
  if (h->buf.env_reward) return fail(MG_EUNSUPPORTED, "mg_rollout / mg_rollout_random have no env_reward output (emit_env_reward = 0)");
  if (!h->have_reset) return fail(MG_ESTATE, "mg_reset must be called before mg_rollout / mg_rollout_random");
  if (h->dynamic && h->dyn_roll_fn[dyn_dec_index(h)] && !debug_env("MG_DEBUG_DYN_ROLL_STEPS")) {
    // Mutable-grid kinds: the fused kernel of mg_dyn.h (a chunk's grids stay in LDS over the n_steps steps)
    int rc = launch(h, OP_STEP, actions, nullptr, nullptr, nullptr, nullptr, actions ? 0 : 1, nullptr, nullptr, nullptr, nullptr, n_steps, out);
    if (rc == MG_OK) h->t += (uint64_t)n_steps;
    return rc;
  }
  if (h->dynamic) {
    // MG_DEBUG_DYN_ROLL_STEPS (tests / A-B timing: the round-2 form): the same n_steps steps as n_steps
    // launches of the step kernel, each writing straight into its trajectory slot (launch()'s `dest`); the last one writes the
    // handle's own buffers, which are then copied into the last slot (same stream: one full-size obs / reward / dir-done copy
    // per call, the price of leaving the handle exactly where n_steps single calls would).  Same contract, one launch per step.
    const mg_config& c = h->cfg;
    const size_t S = (size_t)h->stride, A = (size_t)c.n_agents;
    const size_t obs_b = h->full_obs ? S * c.width * c.height * 3 : A * S * c.view_size * c.view_size * 3;
    const size_t rew_b = A * S * (c.reward_dtype == MG_REWARD_F32 ? 4 : 8), dd_b = S * h->dd_bytes, act_b = S * MG_SLOTS(c.n_agents);
    int rc = MG_OK;
    for (int32_t k = 0; k < n_steps && rc == MG_OK; ++k) {
      mg_rollout_out slot;
      memset(&slot, 0, sizeof(slot));
      slot.obs_image = out->obs_image + k * obs_b;
      slot.reward = reinterpret_cast<double*>(reinterpret_cast<uint8_t*>(out->reward) + k * rew_b);
      slot.dir_done = out->dir_done + k * dd_b;
      const uint8_t* act_k = actions ? static_cast<const uint8_t*>(actions) + k * act_b : nullptr;
      uint8_t* act_out = (!actions && out->actions) ? out->actions + k * act_b : nullptr;
      rc = launch(h, OP_STEP, act_k, nullptr, nullptr, nullptr, nullptr, actions ? 0 : 1, act_out, nullptr, nullptr, nullptr, 0, nullptr,
                  k < n_steps - 1 ? &slot : nullptr);
      if (rc == MG_OK) h->t += 1;
    }
    if (rc != MG_OK) return rc;
    DeviceGuard g(h->device);
    if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
    const size_t last = (size_t)n_steps - 1;
    HIP_TRY(hipMemcpyAsync(out->obs_image + last * obs_b, h->full_obs ? h->obs_full : h->buf.obs_image, obs_b, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(reinterpret_cast<uint8_t*>(out->reward) + last * rew_b, h->buf.reward, rew_b, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(out->dir_done + last * dd_b, h->buf.dir_done, dd_b, hipMemcpyDeviceToDevice, h->stream));
    return MG_OK;
  }
  int rc = launch(h, OP_STEP, actions, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, n_steps, out);
  if (rc == MG_OK) h->t += (uint64_t)n_steps;
  return rc;
}

int mg_enable_episode_stats(mg_handle* h, double touch_threshold, mg_episode_stats* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  if (h->cfg.n_agents > 5) return fail(MG_EUNSUPPORTED, "episode statistics cover up to five agents");
  DeviceGuard g(h->device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  if (!h->ep.ep_return) {
    const size_t S = (size_t)h->stride, A = (size_t)h->cfg.n_agents;
    int rc = dalloc(h, &h->ep.ep_return, A * S);
    if (rc == MG_OK) rc = dalloc(h, &h->ep.last_return, A * S);
    if (rc == MG_OK) rc = dalloc(h, &h->ep.ep_goal_touches, S);
    if (rc == MG_OK) rc = dalloc(h, &h->ep.ep_len, S);
    if (rc == MG_OK) rc = dalloc(h, &h->ep.last_goal_touches, S);
    if (rc == MG_OK) rc = dalloc(h, &h->ep.last_len, S);
    if (rc == MG_OK) rc = dalloc(h, &h->ep.ep_count, S);
    if (rc != MG_OK) { memset(&h->ep, 0, sizeof(h->ep)); return rc; }
  }
  h->ep_on = true; h->ep_thr = touch_threshold;
  out->ep_return = h->ep.ep_return; out->ep_goal_touches = h->ep.ep_goal_touches; out->ep_len = h->ep.ep_len;
  out->last_return = h->ep.last_return; out->last_goal_touches = h->ep.last_goal_touches; out->last_len = h->ep.last_len;
  out->ep_count = h->ep.ep_count;
  return MG_OK;
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

int mg_get_reset_index(mg_handle* h, uint64_t* out) {
  if (!h || !out) return fail(MG_EINVAL, "null argument");
  *out = h->reset_epoch;
  return MG_OK;
}

int mg_set_reset_index(mg_handle* h, uint64_t n) {
  if (!h) return fail(MG_EINVAL, "null handle");
  if (n >> 63) return fail(MG_EINVAL, "reset index must be below 2^63");
  h->reset_epoch = n;
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

// Which kernel path the handle runs (MG_PATH_GENERIC or MG_PATH_TABLE) and its launch geometry.
int mg_get_launch_info(mg_handle* h, int32_t* path, int32_t* block_threads, int32_t* grid_blocks, int32_t* lds_bytes) {
  if (!h) return fail(MG_EINVAL, "null handle");
  int nw; unsigned grid; size_t lds;
  const void* split_fn = nullptr;
  // the geometry of a plain mg_step launch: a small batch runs the agents-over-waves kernels (step_split_plan); otherwise, on a
  // mutable-grid handle without the env_reward output, the FREE instantiation
  if (step_split_plan(h, &split_fn, &nw, &grid, &lds)) {}
  else if (h->dynamic) dyn_geometry(h, &nw, &grid, &lds, (h->dyn_free_fn[dyn_dec_index(h)] && !h->buf.env_reward) ? 2 : 0);
  else launch_geometry(h, launch_uses_table(h), &nw, &grid, &lds);
  if (path) *path = launch_uses_table(h) ? MG_PATH_TABLE : MG_PATH_GENERIC;
  if (block_threads) *block_threads = nw * kWave;
  if (grid_blocks) *grid_blocks = (int32_t)grid;
  if (lds_bytes) *lds_bytes = (int32_t)lds;
  return MG_OK;
}

__global__ void reward_probe(int max_steps, double* out) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s <= max_steps; s += gridDim.x * blockDim.x)
    out[s] = device_reward((uint32_t)s, max_steps);
}

// Test hook: the reward formula as the device evaluates it.
int mg_debug_device_rewards(int device, int32_t max_steps, double* out) {
  if (!out || max_steps < 1) return fail(MG_EINVAL, "bad argument");
  DeviceGuard g(device);
  if (!g.ok) return fail(MG_EHIP, "hipSetDevice failed");
  double* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, ((size_t)max_steps + 1) * 8));
  hipLaunchKernelGGL(reward_probe, dim3(64), dim3(256), 0, 0, (int)max_steps, d);
  hipError_t e = hipMemcpy(out, d, ((size_t)max_steps + 1) * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MG_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  return MG_OK;
}

#ifdef MG_PHASE_TIMERS
// timing-only builds: the phase counters of mg_dyn.h (sum over all waves, shader-clock cycles), optionally cleared afterwards
int mg_debug_phase_counters(unsigned long long* out16, int reset) {
  if (!out16) return fail(MG_EINVAL, "null argument");
  HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned int> w((size_t)kPhaseWaves * kPhases);
  HIP_TRY(hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_phase), w.size() * sizeof(unsigned int)));
  for (int i = 0; i < 16; ++i) out16[i] = 0;
  for (int k = 0; k < kPhaseWaves; ++k)
    for (int i = 0; i < kPhases; ++i) out16[i] += w[(size_t)k * kPhases + i];
  if (reset) {
    std::fill(w.begin(), w.end(), 0u);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), w.data(), w.size() * sizeof(unsigned int)));
  }
  return MG_OK;
}
#endif

// Test hook (host only, no GPU): the mutable grids' cell code of a public kind id (MG_KIND_ID), the id it exports back
// as, its table word (encode() triple | flags<<24) and the triple the arithmetic decode of the views extracts from the code
// (type | colour<<8; state is always 0 there: a closed or locked door is shown through the overlay instead).
int mg_debug_cell_code(int32_t kind, uint32_t* code, uint32_t* kind_back, uint32_t* table_word, uint32_t* decoded) {
  if (kind < 0 || kind > 255 || !code || !kind_back || !table_word || !decoded) return fail(MG_EINVAL, "bad argument");
  const uint32_t k = code_from_kind((uint32_t)kind);
  *code = k; *kind_back = kind_from_code(k); *table_word = kind_word(k);
  const uint32_t w = (k * kDecMul) & kDecMask;
  *decoded = ((w >> 8) & 255u) | ((w >> 24) << 8);
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
#endif  // MG_TU_DYNROLL
