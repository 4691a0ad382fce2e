// mg_dyn.h — the mutable-grid kernel: every env kind whose grid differs between envs —
// DoorKey, Meetup, Cluttered / FourRooms with layouts drawn per env, and fixed layouts that hold keys,
// balls, boxes, doors or floor tiles.  Included by mgrid.hip inside its anonymous namespace; shares
// Params, Philox, the agent-order / action draws, the v8 shaper and the obs staging with the
// shared-map kernel.
//
// Every env owns a W*H grid of one-byte cell codes (below) plus one "carry" word
// (byte a = what agent a carries).  HBM keeps the grids chunk-blocked and cell-major: logically
// [chunk][cell y*W+x][64 lanes] bytes, i.e. dword rows of 64 lanes (row k, lane L = lanes 4(L%16).. of cell L/16 + 4k);
// four consecutive rows of a lane are stored side by side ([chunk][k/4][L][k%4] dwords, grid_byte()), so a wavefront
// moves its 64 envs' grids to and from LDS as fully coalesced 16-byte vectors, a quarter of the instructions dword rows took.  In LDS the wave's grids are padded by one ring of grey walls,
// [(H+2)*(W+2) cells][68 bytes: 64 lanes + 4 pad]: a lane's byte address is LINEAR in the cell
// coordinates (cell*68 + lane), so a view cell costs one add + one byte read, cells outside the
// grid are reached by clamping the coordinate onto the ring (Grid.slice pads with the same grey
// Wall, multigrid.py:315) and the 17-dword cell stride spreads the lanes' cells over the banks.
// Agents are not stored in the grid; their cells read 0 and they are overlaid from the packed
// position word, as on the shared-map path.
//
// Reference: envs/gym_multigrid/multigrid.py:815-961 (agent_is_done, move_agent, _forward, _pickup,
// _drop, _toggle), :112-122 (Door.toggle), :997-1035 (gen_obs_grid with the carried object on the
// agent's own cell), multigrid_envs/doorkey.py:34-86 (generator, spawn region).

constexpr uint32_t F_OVERLAP = 1u << 28;   // can_overlap(): floor, open door (goal/lava are F_KILL)
constexpr uint32_t F_PICKUP = 1u << 29;    // can_pickup(): key, ball, box
constexpr uint32_t F_DOOR = 1u << 30;
constexpr uint32_t F_BOX = 1u << 24;

// ---- cell codes ------------------------------------------------------------------------------------
// The grids (HBM and LDS), the reset templates and the carry bytes in registers hold one-byte CELL CODES, a bit-field
// form of the public kind ids (include/mgrid.h MG_KIND_ID, which the C ABI keeps speaking: mg_set_static_map /
// mg_set_maps / mg_export_grid / mg_buffers.carry translate):
//     code = X | T<<1 | C<<5      X = opaque (!see_behind()), T = encode() type (4 bits), C = encode() colour (3 bits)
// so that a view cell's (type, colour, 0) bytes and its opacity bit come straight out of the byte with one 24-bit
// multiply and one mask (dyn_view, DEC) instead of a second, dependent LDS read of a kind -> triple table — the
// read the round-2 ablation priced at 15 % of the DoorKey kernel.  Eight bits hold type, colour and opacity of every
// object; what does not fit is a door's closed / locked distinction, so:
//     nothing (None)        T = 1 (what it encodes as: empty), X = 0                      = 2
//     wall                  T = 2, X = 1;   LavaWall (looks like a grey wall, transparent, deadly): T = 2, C = 5, X = 0
//     floor .. lava         T = type, X = 0
//     door  open            T = 4, X = 0;   closed: T = 4, X = 1;   locked: T = 0, X = 1  (T = 0 is no object's type)
//     agents (emit_full's temporary marks, LDS only)  T >= 12
// Kernels that decode arithmetically (DEC) show a closed or locked door through an overlay, like an agent; layouts
// whose doors cannot be tracked that way run the table-decoding instantiation (DEC = false) of the same code.
__host__ __device__ constexpr uint32_t cell_code(uint32_t T, uint32_t C, uint32_t X) { return X | (T << 1) | (C << 5); }
__host__ __device__ inline uint32_t code_T(uint32_t k) { return (k >> 1) & 15u; }
__host__ __device__ inline uint32_t code_C(uint32_t k) { return (k >> 5) & 7u; }
constexpr uint32_t K_NONE = cell_code(MG_T_EMPTY, 0, 0);       // grid.get(...) is None
constexpr uint32_t K_WALL = cell_code(MG_T_WALL, 5, 1);        // grey wall: wall_rect and Grid.slice's padding
constexpr uint32_t K_GOAL = cell_code(MG_T_GOAL, 1, 0);        // green goal
constexpr uint32_t K_DOOR_YELLOW_LOCKED = cell_code(0, 4, 1);  // Door('yellow', is_locked=True), doorkey.py:60
constexpr uint32_t K_KEY_YELLOW = cell_code(MG_T_KEY, 4, 0);   // Key('yellow'), doorkey.py:65
constexpr uint32_t K_DOOR_RED_LOCKED = cell_code(0, 0, 1);     // Door(color='red', is_locked=True), meetup.py:69
constexpr uint32_t K_LAVA = cell_code(MG_T_LAVA, 0, 0);        // red lava
constexpr uint32_t K_LAVAWALL = cell_code(MG_T_WALL, 5, 0);    // LavaWall (lava_walls.py:25-38)
// an agent marked into the LDS grid while a fully observed image is encoded (emit_full): T = 12 + (i >> 2), X | T's low
// bits = i & 3 ... any 16 codes with T >= 12; here i = 4*agent + dir in bits 0..2 and 5
__host__ __device__ constexpr uint32_t agent_mark(uint32_t i) { return 0x18u | (i & 7u) | ((i >> 3) << 5); }
enum { P_GEN = 4, P_KEY = 5, P_OBJDROP = 6 };

// public kind id (MG_KIND_ID) <-> cell code.  Unknown ids come back as walls, as the import always treated them.
__host__ __device__ inline uint32_t code_from_kind(uint32_t p) {
  if (p == 0u) return K_NONE;
  if (p >= 128u) {
    const uint32_t color = ((p - 128u) >> 2) & 7u, state = p & 3u;
    return state == 0u ? cell_code(MG_T_DOOR, color, 0) : state == 1u ? cell_code(MG_T_DOOR, color, 1) : cell_code(0, color, 1);
  }
  const uint32_t type = p >> 3, color = p & 7u;
  if (type == (uint32_t)MG_T_WALL) return cell_code(MG_T_WALL, color, 1);
  if (type == (uint32_t)MG_T_LAVA && color == 5u) return K_LAVAWALL;
  if (type < (uint32_t)MG_T_WALL || type > (uint32_t)MG_T_LAVA || type == (uint32_t)MG_T_DOOR) return K_WALL;
  return cell_code(type, color, 0);
}
__host__ __device__ inline uint32_t kind_from_code(uint32_t k) {
  const uint32_t X = k & 1u, T = code_T(k), C = code_C(k);
  if (T == 0u) return X ? 128u + C * 4u + 2u : 0u;
  if (T == (uint32_t)MG_T_EMPTY || T > (uint32_t)MG_T_LAVA) return 0u;
  if (T == (uint32_t)MG_T_DOOR) return 128u + C * 4u + X;
  if (T == (uint32_t)MG_T_WALL && !X) return (uint32_t)MG_T_LAVA * 8u + 5u;
  return T * 8u + C;
}
// The carry word (one public kind id per byte; only keys, balls and boxes are ever carried, 0 = nothing) as cell codes
// and back, all bytes at once: public = C | T<<3, code = T<<1 | C<<5.
__device__ __forceinline__ uint32_t carry_to_codes(uint32_t p) { return ((p & 0x07070707u) << 5) | ((p & 0x78787878u) >> 2); }
__device__ __forceinline__ uint32_t carry_to_kinds(uint32_t k) { return ((k >> 5) & 0x07070707u) | ((k << 2) & 0x78787878u); }

// cell code -> encode() triple (type | color<<8 | state<<16) + behaviour flags.  256 entries, built
// once per workgroup into LDS: the action phase reads it for the cell in front of an agent, the fully observed
// images (emit_full) and the table-decoding views (DEC = false) for every cell.  door_state = what a non-open door
// with T = 4 encodes as (1 = closed).
__host__ __device__ inline uint32_t kind_word(uint32_t k) {
  const uint32_t X = k & 1u, T = code_T(k), C = code_C(k);
  if (T >= 12u) {        // emit_full's agent marks
    const uint32_t i = (k & 7u) | ((k >> 5) << 3), id = (i >> 2) & 3u, dir = i & 3u;
    return (uint32_t)MG_T_AGENT | (id << 8) | (dir << 16);        // Agent.encode, multigrid.py:137-139
  }
  if (T == 0u)           // locked door (is_locked: can_overlap / see_behind False), Door.encode state 2
    return X ? ((uint32_t)MG_T_DOOR | (C << 8) | (2u << 16) | F_DOOR | F_OPAQUE) : ((uint32_t)MG_T_EMPTY | F_FREE);
  uint32_t w = T | (C << 8);
  switch (T) {
    case MG_T_EMPTY: w = (uint32_t)MG_T_EMPTY | F_FREE; break;
    case MG_T_WALL: w |= X ? F_OPAQUE : F_KILL; break;        // X = 0: LavaWall (lava_walls.py:25-38), deadly and transparent
    case MG_T_FLOOR: w |= F_OVERLAP; break;
    case MG_T_DOOR: w |= F_DOOR | (X ? (F_OPAQUE | (1u << 16)) : F_OVERLAP); break;   // Door.can_overlap / see_behind = is_open
    case MG_T_KEY: case MG_T_BALL: w |= F_PICKUP; break;
    case MG_T_BOX: w |= F_PICKUP | F_BOX; break;
    case MG_T_GOAL: w |= F_KILL | F_GOAL; break;
    case MG_T_LAVA: w |= F_KILL; break;
    default: break;                                            // a solid, inert object
  }
  return w;
}

// LDS grid of one wave: cell (x, y), x in [-1, W], y in [-1, H] (the ring holds grey walls), lane l
// at byte ((y+1)*(W+2) + x+1)*kCellB + l.  gl = the wave's grid base + lane.
#ifndef MG_VIEW_PHASES
#define MG_VIEW_PHASES 1
#endif
constexpr int kCellB = 68;    // bytes per cell row: 64 lanes + 4 (bank spreading)
constexpr int kCellW = 17;    // the same in dwords
constexpr int kMaxGoals = 6; // MeetupEnv goal doors kept per env: aux words 0..2, two goals (x | y<<6, 0xFFFF = none) per word
constexpr int kPreMax = 16;   // grid dwords per lane fetched per batch when a grid is not prefetched
// PRE (template parameter of the kernels) = grid dwords per lane prefetched in registers: 16 covers grids up to 64
// cells; the view-5 and the fully observed instantiations also exist with 64 (grids up to 16x16 — they have the
// registers for it at two waves per SIMD), picked at mg_create for grids that need it.
// (24-bit multiplies: full rate, where the 32-bit ones issue at a quarter of it; every factor here is below 2^13)
__device__ __forceinline__ int g_off(const Params& p, int x, int y) { return __mul24(__mul24(y + 1, p.W + 2) + x + 1, kCellB); }
__device__ __forceinline__ uint32_t g_get(const uint8_t* gl, int off) { return gl[off]; }
__device__ __forceinline__ void g_set(uint8_t* gl, int off, uint32_t k) { gl[off] = (uint8_t)k; }

// place_obj's acceptance test (multigrid.py:603-613) inside x < xmax on this env's own grid
template <int A>
__device__ __forceinline__ bool dyn_free(const Params& p, const uint8_t* g, pd_t<A> pd, uint32_t occ, int x, int y, int xmax) {
  if ((unsigned)x >= (unsigned)min(xmax, p.W) || (unsigned)y >= (unsigned)p.H) return false;
  if (g_get(g, g_off(p, x, y)) != K_NONE) return false;
  const uint32_t pos = (uint32_t)x | ((uint32_t)y << 6);
  bool ok = true;
#pragma unroll
  for (int b = 0; b < A; ++b) ok = ok && !(((occ >> b) & 1u) && (pd_get(pd, b) & AW_POS) == pos);
  return ok;
}

// place_obj (multigrid.py:560-627) for an agent (is_agent: a direction is drawn too, :665-666) or an
// object, in the rectangle x < xmax.  Injected draw (have_inj, injw = int8x4 x,y,dir,0) or Philox
// rejection sampling keyed (purpose, a, attempt).  Returns x | y<<6 | dir<<12 | MG_ERR_*<<16.
template <int A>
__device__ __forceinline__ uint32_t place_dyn(const Params& p, const uint8_t* g, pd_t<A> pd, uint32_t occ, int a,
                                              int purpose, int xmax, bool have_inj, uint32_t injw, bool is_agent,
                                              uint32_t gid) {
  int x = -1, y = -1, dir = 0;
  uint32_t err = 0;
  if (have_inj) {
    x = (int8_t)(injw & 255u); y = (int8_t)((injw >> 8) & 255u); dir = is_agent ? (int8_t)((injw >> 16) & 255u) : 0;
    if (x < 0) { err = MG_ERR_MISSING_DRAW; x = -1; }
    else if (!dyn_free<A>(p, g, pd, occ, x, y, xmax) || (unsigned)dir > 3u) { err = MG_ERR_BAD_RESPAWN; x = -1; }
  } else {
    const uint32_t nx = (uint32_t)min(xmax, p.W);
    for (int k = 0; k < kMaxAttempts; ++k) {
      const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32),
                          ((uint32_t)purpose << 28) | ((uint32_t)a << 20) | (uint32_t)k, p.seed_lo, p.seed_hi);
      const int cx = (int)__umulhi(r.x, nx), cy = (int)__umulhi(r.y, (uint32_t)p.H);
      if (dyn_free<A>(p, g, pd, occ, cx, cy, xmax)) { x = cx; y = cy; dir = is_agent ? (int)(r.z >> 30) : 0; break; }
    }
  }
  if (x < 0) {   // exit condition of the sampling loop: first free cell in storage order, dir 0
    dir = 0;
    for (int j = 0; j < p.H && x < 0; ++j)
      for (int i = 0; i < p.W && i < xmax; ++i)
        if (dyn_free<A>(p, g, pd, occ, i, j, xmax)) { x = i; y = j; break; }
    if (x < 0) { x = 0; y = 0; }
  }
  return (uint32_t)x | ((uint32_t)y << 6) | ((uint32_t)dir << 12) | (err << 16);
}

// clamp(v, -1, hi) in one instruction (hipcc only forms v_med3_i32 when it can prove lo <= hi)
__device__ __forceinline__ int clamp_ring(int v, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(v), "v"(hi));
  return r;
}

// gen_obs_grid (multigrid.py:997-1035) from this env's own grid: view cell (vx,vy) = world cell
// pos + f*(V-1-vy) + r*(vx-V/2); cells outside the grid are Grid.slice's grey walls (:315), reached
// by clamping each coordinate onto the ring; the own cell shows the carried object (:1029-1033).
// Exactly one of (r.x, f.x) and one of (r.y, f.y) is non-zero, so the clamped coordinates separate:
// offset(vx, vy) = U[vx] + Vv[vy] with 2V precomputed terms.
__device__ __forceinline__ int lane_of(const uint8_t*) { return (int)(threadIdx.x & 63u); }
// DEC: a cell code k = X | T<<1 | C<<5 times kDecMul is k<<7 | k<<19 (the two copies share no bit), which has T in the
// low nibble of byte 1 and C in the low bits of byte 3; kDecMask keeps exactly those, so the product is the cell's
// (state = 0, type, 0, colour) bytes — what pack_view<V, true> picks its stream bytes from.
constexpr uint32_t kDecMul = (1u << 7) | (1u << 19), kDecMask = 0x07000F00u;
template <int V, bool DEC>
__device__ __forceinline__ void dyn_view(const Params& p, const uint8_t* g, const uint32_t* lut, uint32_t me, uint32_t ck,
                                         uint32_t (&s)[ViewGeom<V>::NW + 1], uint64_t& vm) {
  const int x = aw_x(me), y = aw_y(me), d = aw_d(me);
  // vx runs along the agent's right axis and vy against its forward axis; one of the two is world x, the other world
  // y, so each needs one clamped coordinate per step, scaled by that axis' byte stride (x: a cell, y: a padded row)
  const bool fwd_x = (d & 1) == 0;                       // facing east or west
  const int fs = (d == 0 || d == 1) ? 1 : -1;            // DIR_TO_VEC: east (1,0), south (0,1), west (-1,0), north (0,-1)
  const int rs = (d == 1 || d == 2) ? -1 : 1;            // right = (-f.y, f.x): east (0,1), south (-1,0), west (0,-1), north (1,0)
  const int PWB = __mul24(p.W + 2, kCellB);
  const int Sr = fwd_x ? PWB : kCellB, Sf = fwd_x ? kCellB : PWB;
  const int Lr = fwd_x ? p.H : p.W, Lf = fwd_x ? p.W : p.H;
  int r = (fwd_x ? y : x) - rs * (V / 2), f = (fwd_x ? x : y) + fs * (V - 1);   // view cell (0, 0)
  int U[V], Vv[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    U[i] = __mul24(clamp_ring(r, Lr), Sr) + Sr;          // (coordinate + 1) * stride: the ring is row / column 0
    Vv[i] = __mul24(clamp_ring(f, Lf), Sf) + Sf;
    r += rs;
    f -= fs;
  }
  uint32_t code[V * V];
  uint32_t opq[V];
#pragma unroll
  for (int j = 0; j < V; ++j) opq[j] = 0;
  // passes, each over all cells, so that the LDS reads of a view are in flight for many cells at once instead of a
  // handful (two waves per SIMD do not hide the round trips)
#pragma unroll
#if MG_ABL == 6   // timing-only: the same 49 byte gathers, every lane on its own bank (wrong images) — what a conflict-free layout could buy
  for (int c = 0; c < V * V; ++c) code[c] = (g - (lane_of(g)))[(lane_of(g) << 2) + ((U[c / V] + Vv[c % V]) & 0x0F03)];
#else
  for (int c = 0; c < V * V; ++c) code[c] = g_get(g, U[c / V] + Vv[c % V]);
#endif
#if MG_VIEW_PHASES
  __builtin_amdgcn_sched_barrier(0);
#endif
  if constexpr (DEC) {
    // opacity = bit 0 of the code: {code, row} >> 1 shifts it in at the top, so walking vx upwards leaves bit vx of the
    // row at 32 - V + vx; then the (type, colour) bytes by one multiply each — no second LDS read
#pragma unroll
    for (int vx = 0; vx < V; ++vx)
#pragma unroll
      for (int vy = 0; vy < V; ++vy) opq[vy] = __builtin_amdgcn_alignbit(code[vx * V + vy], opq[vy], 1);
#pragma unroll
    for (int vy = 0; vy < V; ++vy) opq[vy] >>= 32 - V;
#pragma unroll
    for (int c = 0; c < V * V; ++c) code[c] = __umul24(code[c], kDecMul);
    // the own cell shows the carried object or None (multigrid.py:1029-1033)
    pack_view<V, true>(code, opq, p.see_through != 0, __umul24(ck ? ck : K_NONE, kDecMul), s, vm);
  } else {
    // the second, dependent LDS read of every cell: code -> its table word
#pragma unroll
#if MG_ABL == 9   // timing-only: no second, dependent LDS read per view cell (wrong images)
    for (int c = 0; c < V * V; ++c) code[c] = code[c] * 0x00010101u;
#else
    for (int c = 0; c < V * V; ++c) code[c] = lut[code[c]];
#endif
#if MG_VIEW_PHASES
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int vx = V - 1; vx >= 0; --vx)
#pragma unroll
      for (int vy = 0; vy < V; ++vy) opq[vy] = view_row_bit(opq[vy], code[vx * V + vy]);
    pack_view<V, false>(code, opq, p.see_through != 0, lut[ck ? ck : K_NONE] & 0xFFFFFFu, s, vm);
  }
}

// Inputs of one chunk, requested one chunk ahead with inline-asm loads and retired with a counted
// s_waitcnt (see load_chunk / wait_chunk in mgrid.hip for why).
// carried kinds, one byte per agent: a dword up to four agents, two dwords (one u64 in the kernel) for five
template <int A> struct CarryOf { typedef uint32_t type; };
template <> struct CarryOf<5> { typedef uint64_t type; };
template <int A> using carry_t = typename CarryOf<A>::type;

template <int A, int PRE>
struct DynIn {
  uint32_t st[A + 1 > 4 ? A + 1 : 4];   // env_state words (A + 1 used)
  u32x4 q;                              // A == 3: the same four words, fetched as one 16-byte vector
  uint32_t carry, act;
  uint32_t carry2, act2;    // five agents only: the second dword of the carry word / of the action slots
  uint32_t aux[4];          // MEETUP: goal positions and past distances
  u32x4 pre4[PRE / 4];      // this lane's share of the chunk's grid rows (rows 4b..4b+3 of lane L), GW <= PRE only
};

// byte of (cell ci, lane l) inside a chunk's grid block (see the layout at the top of this file)
__host__ __device__ inline int64_t grid_byte(int ci, int l) {
  const int D = ci * 16 + (l >> 2), k = D >> 6, L = D & 63;
  return ((((int64_t)(k >> 2) * 64 + L) * 4 + (k & 3)) << 2) + (l & 3);
}

template <int A, int PRE>
__device__ __forceinline__ void dyn_load(const Params& p, int64_t env0, int lane, DynIn<A, PRE>& in) {
  const int64_t n = env0 + lane;
#pragma unroll
  for (int a = 0; a < (A + 1 > 4 ? A + 1 : 4); ++a) in.st[a] = 0u;
  in.q = u32x4{0, 0, 0, 0};
  if constexpr (A == 3) {
    const void* ptr = reinterpret_cast<const uint4*>(p.env_state) + n;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(in.q) : "v"(ptr) : "memory");
  } else {
#pragma unroll
    for (int a = 0; a <= A; ++a) in.st[a] = asm_load_u32(p.env_state + n * (A + 1) + a);
  }
  in.carry = asm_load_u32(p.carry + n * kSlotWords<A>);
  in.act = in.act2 = 0x06060606u;
  in.carry2 = 0u;
  if (A > 4) in.carry2 = asm_load_u32(p.carry + n * kSlotWords<A> + 1);
  if (p.op == OP_STEP && !p.random_actions) {
    in.act = asm_load_u32(p.actions + n * kSlotWords<A>);
    if (A > 4) in.act2 = asm_load_u32(p.actions + n * kSlotWords<A> + 1);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    in.aux[k] = 0u;
    if (p.aux) in.aux[k] = asm_load_u32(p.aux + n * 4 + k);
  }
  const uint4* ghbm = reinterpret_cast<const uint4*>(p.grid) + ((env0 >> 6) * (int64_t)p.GB * kWave + lane);
#pragma unroll
  for (int b = 0; b < PRE / 4; ++b) {
    // every load is issued (a branch around one would leave the compiler a phi to resolve with copies of a register
    // whose load is still in flight); the ones past the grid re-read the chunk's first vector, which the L2 holds
    const void* ptr = (b < p.GB && p.GW <= PRE) ? ghbm + b * kWave : ghbm;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(in.pre4[b]) : "v"(ptr) : "memory");
  }
}

template <int A, int K, int PRE>
__device__ __forceinline__ void dyn_wait(DynIn<A, PRE>& in) {
  static_assert(A <= 5 && PRE % 16 == 0, "operand lists below");
  if constexpr (A == 3) {
    asm volatile("s_waitcnt vmcnt(%7)"
                 : "+v"(in.q), "+v"(in.carry), "+v"(in.act), "+v"(in.aux[0]), "+v"(in.aux[1]), "+v"(in.aux[2]), "+v"(in.aux[3])
                 : "n"(K)
                 : "memory");
    in.st[0] = in.q.x; in.st[1] = in.q.y; in.st[2] = in.q.z; in.st[3] = in.q.w;
  } else if constexpr (A <= 2) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(in.st[0]), "+v"(in.st[1]), "+v"(in.st[2]), "+v"(in.st[3]),
                   "+v"(in.carry), "+v"(in.act), "+v"(in.aux[0]), "+v"(in.aux[1]), "+v"(in.aux[2]), "+v"(in.aux[3])
                 : "n"(K)
                 : "memory");
  } else if constexpr (A == 4) {
    asm volatile("s_waitcnt vmcnt(%11)"
                 : "+v"(in.st[0]), "+v"(in.st[1]), "+v"(in.st[2]), "+v"(in.st[3]), "+v"(in.st[4]),
                   "+v"(in.carry), "+v"(in.act), "+v"(in.aux[0]), "+v"(in.aux[1]), "+v"(in.aux[2]), "+v"(in.aux[3])
                 : "n"(K)
                 : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%14)"
                 : "+v"(in.st[0]), "+v"(in.st[1]), "+v"(in.st[2]), "+v"(in.st[3]), "+v"(in.st[4]), "+v"(in.st[5]),
                   "+v"(in.carry), "+v"(in.carry2), "+v"(in.act), "+v"(in.act2),
                   "+v"(in.aux[0]), "+v"(in.aux[1]), "+v"(in.aux[2]), "+v"(in.aux[3])
                 : "n"(K)
                 : "memory");
  }
  // the grid rows: volatile asm statements keep their order, so no use of these registers can move above the wait
#pragma unroll
  for (int b = 0; b < PRE / 4; b += 4)
    asm volatile("" : "+v"(in.pre4[b]), "+v"(in.pre4[b + 1]), "+v"(in.pre4[b + 2]), "+v"(in.pre4[b + 3]) : : "memory");
}

// MeetupEnv.get_dist (meetup.py:78-84): Manhattan distance of every agent to the goal whose summed
// distance over the agents is smallest (np.argmin: the first such goal).  Returns one byte per agent.
template <int A>
__device__ __forceinline__ uint32_t meetup_dist(const Params& p, pd_t<A> pd, const uint32_t (&goal)[3], uint32_t& sum_out) {
  uint32_t best = 0, best_sum = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < kMaxGoals; ++i) {
    if (i >= p.n_goals) continue;
    const uint32_t gp = goal[i >> 1] >> (16 * (i & 1));              // two goals per aux word
    const int gx = (int)(gp & 63u), gy = (int)((gp >> 6) & 63u);
    uint32_t sum = 0, bytes = 0;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const uint32_t me = pd_get(pd, a);
      const uint32_t dd = (uint32_t)(abs(gx - aw_x(me)) + abs(gy - aw_y(me)));
      sum += dd;
      bytes |= dd << (8 * a);
    }
    if (sum < best_sum) { best_sum = sum; best = bytes; }
  }
  sum_out = best_sum;
  return best;
}

// Fully observed obs (multigrid.py:1043-1045): Grid.encode() of the whole grid, [x][y][3] bytes, with
// the agents in their cells as (10, id, dir).  No staging: the wave's grids already sit in LDS
// cell-major, so the wave gathers the 64 images straight into output order — lane l produces the
// 16-byte vectors l, l+64, ... of the chunk's 64*W*H*3 contiguous bytes and every store instruction
// writes 1 KB of contiguous bytes.  The agents are marked into the LDS grid for the duration (kinds
// agent_mark(4*id + dir)); the next chunk's copy-in overwrites every interior cell, so they never leak — except in the
// fused rollout, whose grids stay in LDS from step to step: RESTORE puts back what the marks covered.
template <int A, bool RESTORE = false>
__device__ __forceinline__ void emit_full(const Params& p, uint8_t* g, const uint8_t* gwave, const uint32_t* lut,
                                          const uint16_t* cpc, pd_t<A> pd, int lane, uint8_t* gchunk, uint8_t* gchunk2 = nullptr,
                                          int part = 0, int parts = 1) {     // this wave writes vectors part*64 + lane, + 64*parts, ... (split rollout)
  const int cells = p.W * p.H, FB = cells * 3;
  uint32_t under[A];
#pragma unroll
  for (int b = 0; b < A; ++b) {
    const uint32_t o = pd_get(pd, b);
    const int off = g_off(p, aw_x(o), aw_y(o));
    if (RESTORE) under[b] = g_get(g, off);
    g_set(g, off, agent_mark(4u * (uint32_t)b + (uint32_t)aw_d(o)));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // lane l produces the 16-byte vectors l, l+64, ... of the chunk; 16 bytes starting at byte r0 of an
  // image span exactly six consecutive cells (3*c0 + m .. 3*c0 + m + 15, m = r0 % 3), the last of
  // which may belong to the next image
  uint4* dst = reinterpret_cast<uint4*>(gchunk);
  uint4* dst2 = reinterpret_cast<uint4*>(gchunk2);   // wave-uniform; only the fused rollout's last step passes one
  const int nv = 4 * FB;                        // 16-byte vectors in the 64-image chunk
  const int vs = kWave * parts;                 // distance between two vectors of a lane
  int e = 0, r0 = 16 * (lane + kWave * part);   // image index within the chunk and byte offset within it
  constexpr int UF = 2;                         // vectors per lane in flight (the dependent LDS reads of each overlap)
  for (int q = lane + kWave * part; q < nv; q += UF * vs) {
    uint32_t k[UF][6], m[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      while (r0 >= FB) { r0 -= FB; ++e; }
      const bool live = q + u * vs < nv;
      const int c0 = live ? (int)__umulhi((uint32_t)r0, 0x55555556u) : 0;   // r0 / 3
      m[u] = live ? (uint32_t)(r0 - 3 * c0) : 0u;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        int c = c0 + j, ee = live ? e : 0;
        if (c >= cells) { c -= cells; ee += 1; }       // never past image 63: the chunk's last vector ends on its last byte
        k[u][j] = gwave[(int)cpc[c] * kCellB + ee];
      }
      r0 += 16 * vs;
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      uint32_t w[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) w[j] = lut[k[u][j]] & 0xFFFFFFu;
      // the six 3-byte codes as a byte stream, then 16 bytes from offset m
      const uint32_t s0 = w[0] | (w[1] << 24), s1 = (w[1] >> 8) | (w[2] << 16), s2 = (w[2] >> 16) | (w[3] << 8);
      const uint32_t s3 = w[4] | (w[5] << 24), s4 = w[5] >> 8;
      uint4 v;
      v.x = __builtin_amdgcn_alignbyte(s1, s0, m[u]);
      v.y = __builtin_amdgcn_alignbyte(s2, s1, m[u]);
      v.z = __builtin_amdgcn_alignbyte(s3, s2, m[u]);
      v.w = __builtin_amdgcn_alignbyte(s4, s3, m[u]);
      if (q + u * vs < nv) {
        store_vec16(dst + q + u * vs, v, p.nt_obs != 0);
        if (dst2) store_vec16(dst2 + q + u * vs, v, false);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (RESTORE) {                                 // (agents never share a cell, so the order does not matter)
#pragma unroll
    for (int b = 0; b < A; ++b) {
      const uint32_t o = pd_get(pd, b);
      g_set(g, g_off(p, aw_x(o), aw_y(o)), under[b]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// -DMG_PHASE_TIMERS (timing-only builds, tools/gpu.sh build ... ; never the product): where a wave's time goes.  The shader
// clock (s_memtime) is read between the phases of a chunk and the differences are summed over all waves into g_phase[],
// which mg_debug_phase_counters hands to tools/kbench.  Wall cycles of the wave, stalls included — the phase breakdown
// rocprofv3's PC sampling would give, which this pool does not offer.
#ifdef MG_PHASE_TIMERS
constexpr int kPhaseWaves = 8192, kPhases = 12;      // one row of counters per wave of the launch: no atomics, nothing shared
static __device__ unsigned int g_phase[kPhaseWaves * kPhases];
#define MG_PH_PARAM , uint32_t (&ph_acc)[kPhases]
#define MG_PH_ARG , ph_acc
#define MG_PH_BEGIN() uint64_t ph_t = __builtin_amdgcn_s_memtime()
#define MG_PH(i) do { const uint64_t ph_now = __builtin_amdgcn_s_memtime(); ph_acc[i] += (uint32_t)(ph_now - ph_t); ph_t = ph_now; } while (0)
#else
#define MG_PH_PARAM
#define MG_PH_ARG
#define MG_PH_BEGIN() do {} while (0)
#define MG_PH(i) do {} while (0)
#endif

// Fused K-step rollout on mutable grids (mg_dyn_rollout_kernel): what a chunk keeps in registers from one step of the
// launch to the next — its 64 grids stay in the wave's LDS meanwhile — and what tells a step where it stands.
template <int A> struct DynSt {
  pd_t<A> pd; uint32_t shw[A]; uint32_t st; carry_t<A> carry; uint32_t goal[3]; uint32_t past; bool dirty;
};
struct DynRoll {
  bool last;                 // the launch's last step: grids and state go back to HBM, and the outputs also go to the handle's own buffers
  bool split; int me;        // small batches (p.roll_split): A waves share the chunk — each runs the whole action phase on its own LDS copy
                             // of the grids and does agent `me`'s shaper, reward, view and image (fully observed: its share of the
                             // image vectors); agent 0's wave also stores dir_done, carry, aux, the step word and the grids
  uint8_t* obs2; double* reward2; uint8_t* dd2;     // those buffers (null unless last)
};

// The start of a chunk: this wave's 64 grids, HBM rows -> padded LDS grid (dword d = lane + 64k of the chunk holds lanes
// 4*(d%16)..+3 of cell d/16; it goes to LDS dword pidx[cell]*17 + d%16), and the chunk's state unpacked from its input words.
template <int A, int PRE>
__device__ __forceinline__ void dyn_begin(const Params& p, uint32_t* gw, int lane, int64_t env0, const DynIn<A, PRE>& in, DynSt<A>& S) {
  const int cells = p.W * p.H;
  const uint4* const ghbm = reinterpret_cast<const uint4*>(p.grid) + (env0 >> 6) * (int64_t)p.GB * kWave;   // this chunk's grid block
  const int l4 = lane & 15, c4 = lane >> 4;
  // LDS dword of cell ci = y*W + x for this lane's four envs: ((y+1)(W+2) + x+1)*17 + l4 = (ci + 2y + W+3)*17 + l4, with
  // y = ci / W as one 24-bit multiply and a shift (inv_w20 = ceil(2^20 / W): exact for ci < 2^20 / W, i.e. any grid here).
  // Arithmetic, not the pidx[] table: a table lookup per row is a dependent LDS round trip per row (round 3 phase timers:
  // a quarter of a Cluttered-15x15 chunk went into this copy).
  const int ldw0 = __mul24(p.W + 3, kCellW) + l4;
  auto lds_dword = [&](int ci) { return __mul24(ci + 2 * (int)(__umul24((uint32_t)ci, (uint32_t)p.inv_w20) >> 20), kCellW) + ldw0; };
  if (p.GW <= PRE) {
    // The addresses are loop-invariant: with 16 rows the compiler keeps them in registers across chunks (free), with 64
    // rows that would spill, so there the lane's first cell passes through an empty asm and they are recomputed per chunk.
    int c4v = c4;
    if (PRE > 16) asm volatile("" : "+v"(c4v));
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int ci = c4v + 4 * k;
      if (ci < cells) gw[lds_dword(ci)] = in.pre4[k >> 2][k & 3];          // (ci < cells <= 4 GW: no row past the grid)
    }
  } else {
    // grids beyond the prefetch registers: rows fetched kPreMax at a time (all loads of a batch in flight together)
    for (int k0 = 0; k0 < p.GW; k0 += kPreMax) {
      uint4 row[kPreMax / 4];
#pragma unroll
      for (int j = 0; j < kPreMax / 4; ++j) row[j] = (k0 / 4 + j < p.GB) ? ghbm[(k0 / 4 + j) * kWave + lane] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < kPreMax; ++j) {
        const int ci = c4 + 4 * (k0 + j);
        const uint4 r = row[j >> 2];
        if (ci < cells) gw[lds_dword(ci)] = (j & 3) == 0 ? r.x : (j & 3) == 1 ? r.y : (j & 3) == 2 ? r.z : r.w;
      }
    }
  }
  // the rows above were written by other lanes than the ones that read them below
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  S.pd = pd_t<A>{};
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const uint32_t w = in.st[a];
    const uint32_t cx = min(w & 63u, (uint32_t)(p.W - 1)), cy = min((w >> 6) & 63u, (uint32_t)(p.H - 1));
    S.pd = pd_set(S.pd, a, cx | (cy << 6) | (w & 0x3000u));
    S.shw[a] = w & ~AW_POSDIR;
  }
  S.st = in.st[A];
  S.carry = carry_to_codes(in.carry);                         // HBM keeps public kind ids (mg_buffers.carry)
  if constexpr (A > 4) S.carry |= (uint64_t)carry_to_codes(in.carry2) << 32;
  S.goal[0] = in.aux[0]; S.goal[1] = in.aux[1]; S.goal[2] = in.aux[2];
  S.past = in.aux[3];
  S.dirty = false;
}

// One chunk (64 envs) of one step()/reset() on mutable grids.  ROLL = a step of the fused rollout: free-running, no injection,
// no prefetch of a next chunk; state in S between the steps.
template <int A, int V, bool FULL, int PRE, bool DEC, bool ROLL = false>
__device__ __forceinline__ void dyn_chunk(const Params& p, const double* dist, const uint32_t* lut, const uint32_t* base,
                                          const uint16_t* pidx, const uint16_t* cpc, uint32_t* stage, uint32_t* gw, int lane,
                                          int64_t env0, const DynIn<A, PRE>& in, DynIn<A, PRE>& nxt, int64_t env_next,
                                          uint32_t& err_acc MG_PH_PARAM, DynSt<A>* S = nullptr, const DynRoll* R = nullptr) {
  constexpr int ROWB = ViewGeom<V>::ROWB, NW = ViewGeom<V>::NW;
  const bool last = !ROLL || R->last;      // compile-time true outside the rollout
  const bool split = ROLL && R->split;     // (wave-uniform) this wave does agent `me`'s share of the chunk only
  const int me = ROLL ? R->me : 0;
  const int64_t n = env0 + lane;
  const bool active = n < p.n_envs;
  const uint32_t gid = (uint32_t)(p.gid_base + (uint64_t)n);
  const uint32_t all = (1u << A) - 1u;
  const int cells = p.W * p.H;
  MG_PH_BEGIN();

  uint4* const ghbm = reinterpret_cast<uint4*>(p.grid) + (env0 >> 6) * (int64_t)p.GB * kWave;   // this chunk's grid block
  const int l4 = lane & 15, c4 = lane >> 4;
  const int ldw0 = __mul24(p.W + 3, kCellW) + l4;              // see dyn_begin
  auto lds_dword = [&](int ci) { return __mul24(ci + 2 * (int)(__umul24((uint32_t)ci, (uint32_t)p.inv_w20) >> 20), kCellW) + ldw0; };
  DynSt<A> S0;
  if constexpr (!ROLL) dyn_begin<A, PRE>(p, gw, lane, env0, in, S0);
  const DynSt<A>& Sin = ROLL ? *S : S0;
  MG_PH(0);      // grid rows -> LDS
  uint8_t* const g = reinterpret_cast<uint8_t*>(gw) + lane;
  bool dirty = Sin.dirty;        // this env's grid changed: the wave's rows go back to HBM only if some lane's did
  auto gs = [&](int off, uint32_t k) { g_set(g, off, k); dirty = true; };

  pd_t<A> pd = Sin.pd;
  uint32_t shw[A];
#pragma unroll
  for (int a = 0; a < A; ++a) shw[a] = Sin.shw[a];
  uint32_t actw = 0, err = 0;
  uint32_t st = Sin.st;
  carry_t<A> carry = Sin.carry;
  uint32_t goal[3] = {Sin.goal[0], Sin.goal[1], Sin.goal[2]};   // MEETUP: goal_pos[2k] | goal_pos[2k+1] << 16, each x | y<<6
  uint32_t past = Sin.past;                                     // MEETUP: past_goal_dist, one byte per agent
  // agents (re)spawn at x < spawn_w: the DoorKey split (doorkey.py:77-78), the whole grid otherwise
  int spawn_w = p.kind != MG_KIND_DOORKEY ? p.W : (int)min((st >> 24) & 63u, (uint32_t)p.W);
  // DoorKey also keeps its door's row (doorkey.py:58-61; the low five bits, split over the word's free bits 21..23 and
  // 30..31): the door stands at (split_idx, door_idx), and a kernel that decodes cell codes arithmetically shows it,
  // while it is closed or locked, through emit_obs' overlay
  uint32_t dk_bits = p.kind == MG_KIND_DOORKEY ? (st & 0xFFE00000u) : 0u;
  uint32_t done_flag = 0;

  const bool is_step = p.op == OP_STEP;
  if (is_step) {
    U4 blk = U4{0u, 0u, 0u, 0u};                                                // the step's one Philox block (mgrid.hip)
    if (MG_ABL != 3 && (p.random_actions || (A > 1 && !p.order))) blk = step_block(p, gid, P_ORDER);
    if (p.random_actions) actw = draw_actions<A>(p, n, gid, active, blk);
    else {
      uint32_t a0 = in.act, a1 = in.act2;
      if constexpr (ROLL) {                                   // this step's slot of the caller's action ring (p.actions points at it)
        a0 = p.actions[n * kSlotWords<A>];
        if (A > 4) a1 = p.actions[n * kSlotWords<A> + 1];
      }
#pragma unroll
      for (int a = 0; a < A; ++a) actw |= min(((a < 4 ? a0 >> (8 * a) : a1 >> (8 * (a - 4)))) & 255u, 7u) << (3 * a);
    }
    uint32_t step = (st & 0xFFFFu) + 1u;                                        // multigrid.py:968
    uint32_t dbits = (st >> 16) & all;
    if (step > 0xFFFFu) step = 0xFFFFu;
    const uint32_t ordw = agent_order<A, false>(p, n, active, err, blk);
    MG_PH(1);    // unpack, actions, order (Philox)

    uint32_t gbits = 0;
#pragma unroll 1
    for (int k = 0; k < A; ++k) {                                               // step_one_agent, multigrid.py:860-904
      const int a = (int)((ordw >> (kOrdBits<A> * k)) & ((1u << kOrdBits<A>) - 1u));
      const uint32_t s = pd_get(pd, a);
      const uint32_t act = (actw >> (3 * a)) & 7u;
      const int x = aw_x(s), y = aw_y(s), d = aw_d(s);
      const int fx = x + (d == 0) - (d == 2), fy = y + (d == 1) - (d == 3);     // front_pos, pre-action
      const uint32_t fpos = (uint32_t)fx | ((uint32_t)fy << 6);
      bool blocked = false;                                                     // another agent in front
#pragma unroll
      for (int b = 0; b < A; ++b) blocked = blocked || (b != a && (pd_get(pd, b) & AW_POS) == fpos);
      // the front cell of an agent on the grid is on the grid or on the ring of walls; nothing below
      // ever writes to a wall cell, so the ring stays intact.  (A map without a border wall makes the
      // reference fault in Grid.get; here the outside simply reads as wall.)
      const int ci = g_off(p, fx, fy);
      const uint32_t kf = g_get(g, ci);
      const uint32_t fl = lut[kf];
      const uint32_t ck = (uint32_t)(carry >> (8 * a)) & 255u;
      const int nd = (act == 0u) ? ((d + 3) & 3) : (act == 1u) ? ((d + 1) & 3) : d;   // :867-876
      uint32_t npd = (s & AW_POS) | ((uint32_t)nd << 12);
      if (act == 2u) {                                                          // _forward :906-924
        if (!blocked) {
          if (fl & F_KILL) {                                                    // agent_is_done :815-832
            dbits |= 1u << a;
            if (fl & F_GOAL) gbits |= 1u << a;
            if (ck) {
              // the carried object is placed anywhere on the grid first (:827-829); the finishing
              // agent's old cell still counts as occupied (agent_pos is cleared only afterwards, :655)
              const bool inj = p.objdrop != nullptr;
              const uint32_t w = place_dyn<A>(p, g, pd, all, a, P_OBJDROP, p.W, inj, inj ? p.objdrop[n * A + a] : 0u, false, gid);
              gs(g_off(p, aw_x(w), aw_y(w)), ck);
              carry &= ~((carry_t<A>)255u << (8 * a));
              err |= w >> 16;
            }
            const bool inj = p.respawn != nullptr;
            const uint32_t w = place_dyn<A>(p, g, pd, all & ~(1u << a), a, P_RESPAWN, spawn_w, inj,
                                            inj ? p.respawn[n * A + a] : 0u, true, gid);
            npd = w & AW_POSDIR;
            err |= w >> 16;
          } else if (fl & (F_FREE | F_OVERLAP)) {                               // move_agent :834-847
            npd = fpos | ((uint32_t)d << 12);
            if (fl & F_OVERLAP) gs(ci, K_NONE);  // the Agent object replaces what was there, for good
          }
        }
      } else if (act == 3u) {                                                   // _pickup :926-938
        if ((fl & F_PICKUP) && ck == 0u) {
          carry |= (carry_t<A>)kf << (8 * a);
          gs(ci, K_NONE);
        }
      } else if (act == 4u) {                                                   // _drop :940-951
        if ((fl & F_FREE) && !blocked && ck != 0u) {
          gs(ci, ck);
          carry &= ~((carry_t<A>)255u << (8 * a));
        }
      } else if (act == 5u) {                                                   // _toggle :953-961
        if (fl & F_DOOR) {                                                      // Door.toggle :112-122
          if (code_T(kf) == 0u) {                                               // locked: opens for a key of its colour, which is kept
            if (code_T(ck) == (uint32_t)MG_T_KEY && code_C(ck) == code_C(kf)) gs(ci, cell_code(MG_T_DOOR, code_C(kf), 0));
          } else {
            gs(ci, kf ^ 1u);                                                    // is_open = not is_open: the X bit
          }
        } else if (fl & F_BOX) {
          gs(ci, K_NONE);                                                 // Box.toggle: replaced by its (empty) contents
        }
      }
      if (act > 6u) err |= MG_ERR_BAD_ACTION;                                   // assert False :902
      pd = pd_set(pd, a, npd);
    }
    MG_PH(2);    // step_one_agent x A
    double rgoal = 0.0;
    if (gbits) rgoal = device_reward(step, p.max_steps);                        // _reward() :883
    done_flag = (p.competitive && dbits != 0u) || (step >= (uint32_t)p.max_steps);   // :986-993

    double rew[A], out_rew[A];
#pragma unroll
    for (int a = 0; a < A; ++a) rew[a] = ((gbits >> a) & 1u) ? rgoal : 0.0;
    if (p.kind == MG_KIND_MEETUP) {                                             // MeetupEnv.step, meetup.py:86-95
      uint32_t sum;
      const uint32_t cur = meetup_dist<A>(p, pd, goal, sum);
      const bool met = sum == (uint32_t)A;
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const int delta = (int)((past >> (8 * a)) & 255u) - (int)((cur >> (8 * a)) & 255u);
        rew[a] = (double)(delta + (met ? 1 : 0));
      }
      if (met) done_flag = 1u;
      past = cur;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
      out_rew[a] = rew[a];
      if (split && a != me) continue;
      if (p.shaping != MG_SHAPING_NONE && MG_ABL != 4) {
        // get_goal_position (v8:159-169): the per-env scattered goal of a Cluttered layout, else the handle's
        const bool own = (p.kind == MG_KIND_CLUTTERED && p.n_goals != 0) || p.kind == MG_KIND_FOURROOMS || p.grid0 != nullptr;
        const int gx = own ? (int)(goal[0] & 63u) : p.gx, gy = own ? (int)((goal[0] >> 6) & 63u) : p.gy;
        shw[a] = v8_shape<true>(p, dist, shw[a], pd_get(pd, a), (actw >> (3 * a)) & 7u, rew[a], out_rew[a], gx, gy);
      }
    }
    st = step | (dbits << 16) | dk_bits;
    if (active) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        if (split && a != me) continue;
        const int64_t i = (int64_t)a * p.stride + n;
        if (p.reward_f32) reinterpret_cast<float*>(p.reward)[i] = (float)out_rew[a]; else p.reward[i] = out_rew[a];
        if (p.env_reward) p.env_reward[i] = rew[a];
        if (ROLL && R->reward2) {
          if (p.reward_f32) reinterpret_cast<float*>(R->reward2)[i] = (float)out_rew[a]; else R->reward2[i] = out_rew[a];
        }
      }
    }
  }

  MG_PH(3);      // rewards, shaper, reward stores
  // ---- reset() (multigrid.py:466-501) / auto-reset: _gen_grid on this env's own grid ------------
  const bool do_reset = (!is_step) ? (p.mask ? (p.mask[active ? n : 0] != 0) : true) : (done_flag && p.auto_reset);
  if (do_reset) {
    if (p.grid0) {                         // mg_set_maps: this env's own layout (a lane's byte of every cell row)
      const uint8_t* t0 = reinterpret_cast<const uint8_t*>(p.grid0) + (env0 >> 6) * (int64_t)p.GB * (kWave * 16);
      for (int ci = 0; ci < cells; ++ci) gs((int)pidx[ci] * kCellB, t0[grid_byte(ci, lane)]);
    } else {
      for (int ci = 0; ci < cells; ++ci) gs((int)pidx[ci] * kCellB, (base[ci >> 2] >> (8 * (ci & 3))) & 255u);
    }
    carry = 0;                                                                  // multigrid.py:490
    int door = 1;
    const bool inj_gen = p.reset_gen != nullptr;
    const uint32_t genw = inj_gen ? p.reset_gen[n] : 0u;
    if (p.kind == MG_KIND_DOORKEY) {                                            // doorkey.py:46-52
      const int start = p.W <= 5 ? 2 : 3;
      int split;
      if (inj_gen) { split = (int8_t)(genw & 255u); door = (int8_t)((genw >> 8) & 255u); }
      else {
        const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32), (uint32_t)P_GEN << 28, p.seed_lo, p.seed_hi);
        split = start + (int)__umulhi(r.x, (uint32_t)(p.W - 2 - start));        // _rand_int(start_idx, width-2)
        door = 1 + (int)__umulhi(r.y, (uint32_t)(p.W - 3));                     // _rand_int(1, width-2)
      }
      if (split < 1 || split > p.W - 2 || door < 1 || door > p.H - 2) { err |= MG_ERR_BAD_RESPAWN; split = start; door = 1; }
      for (int j = 0; j < p.H; ++j) gs(g_off(p, split, j), K_WALL);       // vert_wall(split_idx, 0)
      spawn_w = split;
      dk_bits = ((uint32_t)split << 24) | (((uint32_t)door & 7u) << 21) | ((((uint32_t)door >> 3) & 3u) << 30);
    }
    if (p.kind == MG_KIND_CLUTTERED || p.kind == MG_KIND_MEETUP) {
      // place_obj for every scattered object in draw order (cluttered.py:28-36, meetup.py:67-72): the
      // goal / goal doors first, then the clutter.  No agent has a position yet (multigrid.py:471-473).
      const uint32_t k1 = p.kind == MG_KIND_MEETUP ? K_DOOR_RED_LOCKED : K_GOAL;
      const uint32_t k2 = p.kind != MG_KIND_CLUTTERED ? K_WALL
                          : (p.gen_flags & MG_GEN_LAVAWALL) ? K_LAVAWALL : (p.gen_flags & MG_GEN_LAVA) ? K_LAVA : K_WALL;
      const int n_obj = p.n_goals + p.n_clutter;
      const bool inj = p.scatter != nullptr;
      goal[0] = goal[1] = goal[2] = 0xFFFFFFFFu;
#pragma unroll 1
      for (int i = 0; i < n_obj; ++i) {
        const uint32_t w = place_dyn<A>(p, g, pd, 0u, i, P_GEN, p.W, inj, inj ? (uint32_t)p.scatter[n * n_obj + i] : 0u, false, gid);
        gs(g_off(p, aw_x(w), aw_y(w)), i < p.n_goals ? k1 : k2);
        err |= w >> 16;
        if (i < p.n_goals) {                       // Meetup goal_pos[i]; Cluttered: the goal the shaper measures to
          const uint32_t sh = 16u * (uint32_t)(i & 1), keep = ~(0xFFFFu << sh), v = (w & AW_POS) << sh;
          if ((i >> 1) == 0) goal[0] = (goal[0] & keep) | v;
          if ((i >> 1) == 1) goal[1] = (goal[1] & keep) | v;
          if ((i >> 1) == 2) goal[2] = (goal[2] & keep) | v;
        }
      }
    }
    if (p.kind == MG_KIND_FOURROOMS) {
      // FourRoomsEnv._gen_grid (fourrooms.py:63-90): walls at x = W/2 and y = H/2 with one gap per segment, drawn
      // in the order top, left, right, bottom; a gap at x <= 1, x >= W-1, y <= 0 or y >= H-1 is not carved
      const int rw = p.W / 2, rh = p.H / 2;
      int g0, g1, g2, g3;                                                       // top y, left x, right x, bottom y
      if (p.scatter) {
        const uint32_t w0 = p.scatter[n * 3], w1 = p.scatter[n * 3 + 1];
        g0 = (int8_t)(w0 & 255u); g1 = (int8_t)(w0 >> 8); g2 = (int8_t)(w1 & 255u); g3 = (int8_t)(w1 >> 8);
      } else {
        const U4 r = philox(gid, (uint32_t)p.t, (uint32_t)(p.t >> 32), (uint32_t)P_GEN << 28, p.seed_lo, p.seed_hi);
        g0 = 1 + (int)__umulhi(r.x, (uint32_t)(rh - 1));
        g1 = 1 + (int)__umulhi(r.y, (uint32_t)(rw - 1));
        g2 = rw + 1 + (int)__umulhi(r.z, (uint32_t)(rw - 1));
        g3 = rh + 1 + (int)__umulhi(r.w, (uint32_t)(rh - 1));
      }
      // two_rooms (fourrooms.py:75-83): no horizontal wall and no draw for it; the lower gap is neither drawn nor
      // carved when height < 7
      const bool two = (p.gen_flags & MG_GEN_TWO_ROOMS) != 0, low = !(two && p.H < 7);
      if (two) { g1 = 1; g2 = rw + 1; }
      if (!low) g3 = rh + 1;
      if (g0 < 1 || g0 >= rh || g1 < 1 || g1 >= rw || g2 <= rw || g2 >= 2 * rw || g3 <= rh || g3 >= 2 * rh) {
        err |= MG_ERR_BAD_RESPAWN; g0 = 1; g1 = 1; g2 = rw + 1; g3 = rh + 1;
      }
      for (int y = 0; y < 2 * rh; ++y) gs(g_off(p, rw, y), K_WALL);
      if (!two) for (int x = 0; x < 2 * rw; ++x) gs(g_off(p, x, rh), K_WALL);
      if (g0 < p.H - 1) gs(g_off(p, rw, g0), K_NONE);                               // x = rw is never <= 1 for W >= 6
      if (!two && g1 > 1) gs(g_off(p, g1, rh), K_NONE);
      if (!two && g2 < p.W - 1) gs(g_off(p, g2, rh), K_NONE);
      if (low && g3 < p.H - 1) gs(g_off(p, rw, g3), K_NONE);
    }
#pragma unroll 1
    for (int a = 0; a < A; ++a) {                                               // place_agent, in agent order
      const bool inj = p.reset_spawn != nullptr;
      const uint32_t w = place_dyn<A>(p, g, pd, (1u << a) - 1u, a, P_SPAWN, spawn_w, inj,
                                      inj ? p.reset_spawn[n * A + a] : 0u, true, gid);
      pd = pd_set(pd, a, w & AW_POSDIR);
      err |= w >> 16;
    }
    if (p.kind == MG_KIND_DOORKEY) {                                            // doorkey.py:58-65
      gs(g_off(p, spawn_w, door), K_DOOR_YELLOW_LOCKED);
      const uint32_t w = place_dyn<A>(p, g, pd, all, 0, P_KEY, spawn_w, inj_gen, genw >> 16, false, gid);
      gs(g_off(p, aw_x(w), aw_y(w)), K_KEY_YELLOW);
      err |= w >> 16;
    }
    if (p.kind == MG_KIND_FOURROOMS) {                                          // place_obj(Goal()) after the agents, fourrooms.py:104
      const bool inj = p.scatter != nullptr;
      const uint32_t w = place_dyn<A>(p, g, pd, all, 1, P_GEN, p.W, inj, inj ? (uint32_t)p.scatter[n * 3 + 2] : 0u, false, gid);
      gs(g_off(p, aw_x(w), aw_y(w)), K_GOAL);
      err |= w >> 16;
      goal[0] = (w & AW_POS) | 0xFFFF0000u; goal[1] = goal[2] = 0xFFFFFFFFu;
    }
    if (p.kind == MG_KIND_MEETUP) {                                             // MeetupEnv.reset, meetup.py:59-62
      uint32_t sum;
      past = meetup_dist<A>(p, pd, goal, sum);
    }
#pragma unroll
    for (int a = 0; a < A; ++a) shw[a] = 0u;
    st = dk_bits;
  }

  MG_PH(4);      // reset / auto-reset
  if (active) {
    if (last && split) {               // each wave its own agent's word (only that one's shaper bits are kept up to date here)
#pragma unroll
      for (int a = 0; a < A; ++a)
        if (a == me) p.env_state[n * (A + 1) + a] = shw[a] | pd_get(pd, a);
      if (me == 0) p.env_state[n * (A + 1) + A] = st;
    } else if (last) {
      if constexpr (A == 3) {          // the four words of an env are one aligned 16-byte vector
        reinterpret_cast<uint4*>(p.env_state)[n] = make_uint4(shw[0] | pd_get(pd, 0), shw[1] | pd_get(pd, 1), shw[2] | pd_get(pd, 2), st);
      } else {
#pragma unroll
        for (int a = 0; a < A; ++a) p.env_state[n * (A + 1) + a] = shw[a] | pd_get(pd, a);
        p.env_state[n * (A + 1) + A] = st;
      }
    }
    if (last && (!split || me == 0)) {
      p.carry[n * kSlotWords<A>] = carry_to_kinds((uint32_t)carry);
      if constexpr (A > 4) p.carry[n * kSlotWords<A> + 1] = carry_to_kinds((uint32_t)(carry >> 32));
      if (p.aux) reinterpret_cast<uint4*>(p.aux)[n] = make_uint4(goal[0], goal[1], goal[2], past);
    }
    if (!split || me == 0) {
      uint64_t dd = (uint64_t)done_flag << (8 * A);
#pragma unroll
      for (int a = 0; a < A; ++a) dd |= (uint64_t)aw_d(pd_get(pd, a)) << (8 * a);
      if (A + 1 <= 4) reinterpret_cast<uint32_t*>(p.dir_done)[n] = (uint32_t)dd;
      else reinterpret_cast<uint64_t*>(p.dir_done)[n] = dd;
      if (ROLL && R->dd2) {
        if (A + 1 <= 4) reinterpret_cast<uint32_t*>(R->dd2)[n] = (uint32_t)dd;
        else reinterpret_cast<uint64_t*>(R->dd2)[n] = dd;
      }
    }
    err_acc |= err;
  }
  // ---- grids: LDS -> HBM rows (the reverse of the copy above; padding lanes' bytes are unused) ---
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (last && (!split || me == 0) && __builtin_amdgcn_ballot_w64(dirty) != 0ull) {          // wave-uniform
    if (p.GW <= PRE) {
      // the mirror of the copy-in, unrolled the same way: the same (kept or recomputed) addresses, every LDS read of
      // the grid in flight before the first store
      int c4v = c4;
      if (PRE > 16) asm volatile("" : "+v"(c4v));
#pragma unroll
      for (int b = 0; b < PRE / 4; ++b) {
        if (b < p.GB) {                                       // rows 4b .. 4b+3 of this lane as one 16-byte store
          uint32_t r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ci = c4v + 4 * (4 * b + j);
            r[j] = ci < cells ? gw[lds_dword(ci)] : 0u;
          }
          ghbm[b * kWave + lane] = make_uint4(r[0], r[1], r[2], r[3]);
        }
      }
    } else {
#pragma unroll 2
      for (int b = 0; b < p.GB; ++b) {
        uint32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ci = c4 + 4 * (4 * b + j);
          r[j] = ci < cells ? gw[lds_dword(ci)] : 0u;
        }
        ghbm[b * kWave + lane] = make_uint4(r[0], r[1], r[2], r[3]);
      }
    }
  }

  MG_PH(5);      // state stores, grid rows -> HBM
  // The next chunk's inputs are requested here, ahead of this chunk's burst of obs stores (at the top of the loop they
  // queued behind the previous chunk's burst: round 3 phase timers, 6 % of a DoorKey chunk went into issuing them).
  // env_next < 0: no next chunk (wave-uniform).  Fully observed kinds keep the request at the top of the loop: the
  // number of stores emit_full issues is not a compile-time constant for the counted wait.
  if constexpr (!FULL && !ROLL) {
    if (env_next >= 0) dyn_load<A, PRE>(p, env_next, lane, nxt);
    MG_PH(9);    // prefetch issue
  }
  // ---- gen_obs (multigrid.py:1037-1067) --------------------------------------------------------
#if MG_ABL != 2
  if (FULL) {
    emit_full<A, ROLL>(p, g, reinterpret_cast<const uint8_t*>(gw), lut, cpc, pd, lane, p.obs_full + env0 * (int64_t)(p.W * p.H * 3),
                 (ROLL && R->obs2) ? R->obs2 + env0 * (int64_t)(p.W * p.H * 3) : nullptr, split ? me : 0, split ? A : 1);
  } else {
    // DEC: the one door whose state the cell code cannot show (DoorKey's, at (split_idx, door_idx) for the whole
    // episode) is read back once per step; gone (walked through while open: move_agent overwrites it) or open -> no overlay
    uint32_t doorw = 0xFFFFFFFFu;
    if (DEC && p.kind == MG_KIND_DOORKEY) {
      const int dx = spawn_w, dy = (int)(((dk_bits >> 21) & 7u) | ((dk_bits >> 30) << 3));
      const uint32_t kd = g_get(g, g_off(p, dx, dy)), T = code_T(kd);
      if ((kd & 1u) && (T == 0u || T == (uint32_t)MG_T_DOOR))
        doorw = (uint32_t)dx | ((uint32_t)dy << 6) | ((T == 0u ? 2u : 1u) << 12) | (code_C(kd) << 16);
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
      if (split && a != me) continue;
      uint32_t s[NW + 1];
      uint64_t vm;
#if MG_ABL == 7   // timing-only: no view computation, the staging and the stores of (meaningless) images stay
      for (int i = 0; i <= NW; ++i) s[i] = pd_get(pd, a) + (uint32_t)i;
      vm = ~0ull;
#else
      dyn_view<V, DEC>(p, g, lut, pd_get(pd, a), (uint32_t)(carry >> (8 * a)) & 255u, s, vm);
#endif
      MG_PH(6);  // views
      emit_obs<A, V, DEC>(pd, a, s, vm, stage, lane, p.obs_image + ((int64_t)a * p.stride + env0) * ROWB, ((p.nt_obs >> a) & 1) != 0,
                          (ROLL && R->obs2) ? R->obs2 + ((int64_t)a * p.stride + env0) * ROWB : nullptr, doorw);
      MG_PH(7);  // staging + obs stores
    }
  }
#endif
  MG_PH(8);
  if constexpr (ROLL) {
    S->pd = pd; S->st = st; S->carry = carry; S->past = past; S->dirty = dirty;
#pragma unroll
    for (int a = 0; a < A; ++a) S->shw[a] = shw[a];
    S->goal[0] = goal[0]; S->goal[1] = goal[1]; S->goal[2] = goal[2];
  }
}

// The LDS of a mutable-grid workgroup, carved and filled (both kernels): [stage: nw x 64*ROWB B][grids: nw x (W+2)(H+2)*68 B]
// [distance LUT: LW*LH f64][kind LUT: 256 u32][base: GW u32][pidx: W*H u16][cpc: W*H u16].  Ends with the workgroup barrier.
struct DynLds {
  uint32_t* stage_all; uint32_t* grids; double* dist; uint32_t* lut; uint32_t* base;
  uint16_t* pidx;      // padded cell index of cell y*W + x
  uint16_t* cpc;       // padded cell index of cell x*H + y (encode() order)
  int stage_b, pcells;
};
__device__ __forceinline__ DynLds dyn_lds_init(const Params& p, uint8_t* smem, int stage_b) {
  DynLds L;
  const int nw = blockDim.x >> 6, tid = threadIdx.x;
  const int pcells = (p.W + 2) * (p.H + 2), cells = p.W * p.H;
  L.stage_b = stage_b; L.pcells = pcells;     // stage_b per wave; fully observed images are gathered without staging (0)
  L.stage_all = reinterpret_cast<uint32_t*>(smem);
  L.grids = reinterpret_cast<uint32_t*>(smem + (size_t)nw * stage_b);
  L.dist = reinterpret_cast<double*>(L.grids + (size_t)nw * pcells * kCellW + ((nw * pcells * kCellW) & 1));
  L.lut = reinterpret_cast<uint32_t*>(L.dist + p.LW * p.LH);
  L.base = L.lut + 256;
  L.pidx = reinterpret_cast<uint16_t*>(L.base + p.GW);
  L.cpc = L.pidx + cells;
  for (int i = tid; i < 256; i += blockDim.x) L.lut[i] = kind_word((uint32_t)i);
  for (int i = tid; i < p.GW; i += blockDim.x) L.base[i] = p.base_grid[i];
  for (int i = tid; i < p.LW * p.LH; i += blockDim.x) L.dist[i] = p.dist_lut[i];
  for (int i = tid; i < cells; i += blockDim.x) {
    L.pidx[i] = (uint16_t)((i / p.W + 1) * (p.W + 2) + (i % p.W) + 1);
    L.cpc[i] = (uint16_t)((i % p.H + 1) * (p.W + 2) + (i / p.H) + 1);
  }
  // every wave's padded grid starts as all grey wall: the ring is never written again
  for (int i = tid; i < nw * pcells * kCellW; i += blockDim.x) L.grids[i] = K_WALL * 0x01010101u;
  __syncthreads();
  return L;
}

// FREE (round 3, second half) = the free-running step as an instantiation of its own: a launch with no injected draws, no reset
// mask and no env_reward output — every mg_step / mg_step_random of a handle created without emit_env_reward — has those
// switches as compile-time constants, so their branches and the scalar registers that carried the pointers go.  The general
// instantiation keeps running resets, replay and env_reward handles.  (These kernels live in mgrid_dynroll.hip's code object.)
template <bool FREE>
__device__ __forceinline__ Params dyn_params(const Params& p0) {
  if constexpr (!FREE) return p0;
  else {
    Params q = p0;
    q.op = OP_STEP; q.order = nullptr; q.respawn = nullptr; q.reset_spawn = nullptr; q.objdrop = nullptr; q.reset_gen = nullptr;
    q.scatter = nullptr; q.mask = nullptr; q.env_reward = nullptr;
    return q;
  }
}

// (Fully observed FREE kernels need 148 VGPRs; forced to 128 for a fourth wave per SIMD — 10 spilled registers, lint clean with
//  16 prefetched rows — Meetup-8x8 ran 107.8 -> 111.0 us at 1M envs and 30.5 -> 33.6 at 131072: three waves stay.)
template <int A, int V, bool FULL, int PRE, bool DEC, bool FREE = false>
__global__ __launch_bounds__(512) void mg_dyn_kernel(const Params p0) {   // up to 8 waves per workgroup (dyn_geometry)
  const Params p = dyn_params<FREE>(p0);
  constexpr int ROWB = ViewGeom<V>::ROWB;
  // (The ~85 dwords of Params do not fit the 102 SGPRs next to the loop's own uniform values: the compiler spills ~350
  //  of them to VGPR lanes and reads whole 16-dword kernarg tuples back, ~700 v_readlane per chunk.  Holding the eight
  //  per-chunk pointers in vector registers instead removed a quarter of those and made DoorKey-8x8 5 % SLOWER
  //  (138 -> 144 us, round 3), so the spills stay.)
  extern __shared__ __align__(16) uint8_t smem[];
  const int nw = blockDim.x >> 6;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;

  // persistent: each wave strides over the 64-env chunks; the first chunk's inputs are requested
  // before the LDS tables are filled
  const int64_t cstride = (int64_t)gridDim.x * nw;
  int64_t c = (int64_t)blockIdx.x * nw + wv;
  DynIn<A, PRE> cur = {}, nxt;
  if (c < p.n_chunks) dyn_load<A, PRE>(p, c * kWave, lane, cur);

  const DynLds L = dyn_lds_init(p, smem, FULL ? 0 : kWave * ROWB);
  const double* dist = L.dist;
  const uint32_t* lut = L.lut; const uint32_t* base = L.base;
  const uint16_t* pidx = L.pidx; const uint16_t* cpc = L.cpc;

  uint32_t* stage = L.stage_all + wv * (L.stage_b / 4);
  uint32_t* gw = L.grids + (size_t)wv * L.pcells * kCellW;
  uint32_t err = 0;
  // vector-memory operations a chunk issues after the next chunk's prefetch, all of them for at least one lane.
  // Partial views: the obs stores (the request is issued right before them, dyn_chunk).  Fully observed: the request is
  // at the top of the loop, so env_state, carry, dir_done (rewards in a step, dirty grid rows and the images on top).
#if MG_ABL == 0 || MG_ABL >= 3
  constexpr int kObs = A * ((4 * ROWB + kWave - 1) / kWave);
#else
  constexpr int kObs = 0;
#endif
  constexpr int kState = A == 3 ? 1 : A + 1;                       // env_state: one 16-byte store with three agents, else a word each
  constexpr int kLo = kObs < 63 ? kObs : 63;                       // partial views: the request sits right before the obs stores
  constexpr int kFull = kState + 2;                                // fully observed: at least these
  dyn_wait<A, 0, PRE>(cur);      // on every path, right after the barrier (see mg_kernel)
#ifdef MG_PHASE_TIMERS
  uint32_t ph_acc[kPhases] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t ph_k = __builtin_amdgcn_s_memtime();
#endif
  for (; c < p.n_chunks; c += cstride) {
    const int64_t cn = c + cstride;
    const bool more = cn < p.n_chunks;          // wave-uniform
    if (FULL && more) dyn_load<A, PRE>(p, cn * kWave, lane, nxt);
#ifdef MG_PHASE_TIMERS
    { const uint64_t now = __builtin_amdgcn_s_memtime(); ph_acc[9] += (uint32_t)(now - ph_k); ph_k = now; }     // prefetch issue
#endif
    dyn_chunk<A, V, FULL, PRE, DEC>(p, dist, lut, base, pidx, cpc, stage, gw, lane, c * kWave, cur, nxt, more ? cn * kWave : -1, err MG_PH_ARG);
#ifdef MG_PHASE_TIMERS
    ph_k = __builtin_amdgcn_s_memtime();
#endif
    if (more) {
      if (FULL) dyn_wait<A, kFull, PRE>(nxt); else dyn_wait<A, kLo, PRE>(nxt);
      cur = nxt;
    }
#ifdef MG_PHASE_TIMERS
    { const uint64_t now = __builtin_amdgcn_s_memtime(); ph_acc[10] += (uint32_t)(now - ph_k); ph_k = now; ph_acc[11] += 1u; }   // counted wait; chunks
#endif
  }
#ifdef MG_PHASE_TIMERS
  { const int wid = (int)(blockIdx.x * nw + wv);
    if (lane == 0 && wid < kPhaseWaves) for (int i = 0; i < kPhases; ++i) g_phase[wid * kPhases + i] += ph_acc[i]; }
#endif
  if (err) atomicOr(p.err, err);
}

#ifndef MG_DYN_ROLL_FULL_WAVES
#define MG_DYN_ROLL_FULL_WAVES 3
#endif
// Fused K-step rollout on mutable grids (mg_rollout / mg_rollout_random on DoorKey, Meetup, drawn layouts, per-env maps):
// p.n_roll consecutive free-running steps in ONE launch — exactly what p.n_roll launches of mg_dyn_kernel compute.  A wave
// takes a chunk through all K steps before its next chunk: the 64 grids are copied into the wave's LDS once and written
// back once (if any step changed them), env_state / carry / aux are read and written once, and step k's obs / rewards /
// dir-done / actions go to slot k of the caller's trajectory buffers (the last step's also to the handle's own buffers, so
// that the handle ends where K single calls would leave it).  What this removes from a step: the launch ramp and drain
// (all of a step's cost at a few thousand envs), the grid and state traffic, and the per-chunk prologue.
template <int A, int V, bool FULL, int PRE, bool DEC>
__global__ __launch_bounds__(512, FULL ? MG_DYN_ROLL_FULL_WAVES : 1) void mg_dyn_rollout_kernel(const Params p) {   // fully observed: three waves per SIMD, like their step kernels
  constexpr int ROWB = ViewGeom<V>::ROWB;
  extern __shared__ __align__(16) uint8_t smem[];
  const int nw = blockDim.x >> 6, cells = p.W * p.H;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  const DynLds L = dyn_lds_init(p, smem, FULL ? 0 : kWave * ROWB);
  const double* dist = L.dist;
  const uint32_t* lut = L.lut; const uint32_t* base = L.base;
  const uint16_t* pidx = L.pidx; const uint16_t* cpc = L.cpc;

  uint32_t* stage = L.stage_all + wv * (L.stage_b / 4);
  uint32_t* gw = L.grids + (size_t)wv * L.pcells * kCellW;
  uint32_t err = 0;
  const int64_t slot_obs = FULL ? p.stride * (int64_t)(cells * 3) : (int64_t)A * p.stride * ROWB;
  const int64_t slot_rew = (int64_t)A * p.stride * (p.reward_f32 ? 4 : 8);
  const int64_t slot_dd = p.stride * p.dd_bytes, slot_act = p.stride * kSlotWords<A>;
  // what a rollout launch never has (launch() passes none of it) as compile-time constants, as in the FREE step kernel: the
  // injection paths, the reset mask and the env_reward output fold away, and with them the scalar registers that carried
  // their pointers (DoorKey-8x8 16.4 -> 15.6 us per step at 131072 envs, Cluttered-15x15 17.2 -> 16.4)
  Params q = dyn_params<true>(p);
  // small batches (p.roll_split): a chunk's agents dealt out over A waves, which each copy its grids into their own LDS and step
  // them identically (same draws, same arithmetic), see DynRoll.  Those A waves sit in one workgroup (launch(): nw is a multiple
  // of A then) and meet at a barrier between reading the chunk's state and the K steps, at whose end wave 0 writes it back: no
  // wave can read what another has already advanced.  The trip count is the same for all waves of a workgroup; a wave without
  // a chunk only keeps the barriers company.
  DynRoll R;
  R.split = p.roll_split != 0;
  const int cpb = R.split ? nw / A : nw;
  R.me = R.split ? wv % A : 0;
  const int64_t n_groups = (p.n_chunks + cpb - 1) / cpb;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t c = g * cpb + (R.split ? wv / A : wv);
    const bool valid = c < p.n_chunks;
    DynIn<A, PRE> cur = {};
    DynSt<A> S;
    if (valid) {
      dyn_load<A, PRE>(p, c * kWave, lane, cur);
      dyn_wait<A, 0, PRE>(cur);            // once per K steps: nothing is in flight across code here
      dyn_begin<A, PRE>(p, gw, lane, c * kWave, cur, S);
    }
    if (R.split) __syncthreads();          // (uniform over the workgroup)
    if (!valid) continue;
#pragma unroll 1
    for (int k = 0; k < p.n_roll; ++k) {
      R.last = k == p.n_roll - 1;
      R.obs2 = R.last ? (FULL ? p.obs_full : p.obs_image) : nullptr;
      R.reward2 = R.last ? p.reward : nullptr;
      R.dd2 = R.last ? p.dir_done : nullptr;
      q.t = p.t + (uint64_t)k;
      if (FULL) q.obs_full = p.traj_obs + k * slot_obs; else q.obs_image = p.traj_obs + k * slot_obs;
      q.reward = reinterpret_cast<double*>(reinterpret_cast<char*>(p.traj_reward) + k * slot_rew);
      q.dir_done = p.traj_dd + k * slot_dd;
      q.actions_out = (p.traj_actions && R.me == 0) ? p.traj_actions + k * slot_act : nullptr;
      q.actions = p.actions ? p.actions + k * slot_act : nullptr;
#ifdef MG_PHASE_TIMERS
      uint32_t ph_acc[kPhases] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
      dyn_chunk<A, V, FULL, PRE, DEC, true>(q, dist, lut, base, pidx, cpc, stage, gw, lane, c * kWave, cur, cur, -1, err MG_PH_ARG, &S, &R);
    }
  }
  if (err) atomicOr(p.err, err);
}

// mg_export_grid: de-block the per-env grids into plain [n][W*H] byte arrays of public kind ids
__global__ void export_grid_kernel(const uint8_t* grid, int GB, int cells, int64_t n_envs, uint8_t* out) {
  const int64_t total = n_envs * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / cells;
    const int ci = (int)(i - n * cells);
    out[i] = (uint8_t)kind_from_code(grid[(n >> 6) * (int64_t)GB * (kWave * 16) + grid_byte(ci, (int)(n & 63))]);
  }
}
