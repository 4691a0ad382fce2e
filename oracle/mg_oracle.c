/*
 * mg_oracle.c — CPU restatement of the reference's MultiGrid step path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product (multigrid_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for everything except MiniGridEnv._reward.
 *   The reference (cxh42/multigrid) has no tests and no golden vectors, and its cell-level
 *   arithmetic lives in gym-minigrid==1.0.3 / gym==0.23.1 (requirements.txt:3-4), which are
 *   not installed here and cannot be fetched.  The multi-agent logic below follows the
 *   reference's own files line by line (citations per function); the minigrid pieces
 *   (Grid.get/set/encode/process_vis, object flags, DIR_TO_VEC, _reward) restate the
 *   published 1.0.3 algorithm as recorded in SURVEY.md Appendix A.  The one data pin the
 *   reference tree holds — env rewards logged in v1_final_results_*.json — is checked in
 *   tests/test_oracle_pins.py.
 *
 * Deliberately literal: an object grid, Grid.slice, rotate_left applied dir+1 times, the
 * two-pass process_vis loops and encode are all executed as the reference executes them, so
 * that the HIP kernels' closed forms (view->world index, bitmask visibility) are checked
 * against the original algorithm and not against themselves.
 *
 * All paths are relative to /root/reference/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define MAXW 64
#define MAXA 8
#define MAXV 9

enum { T_NONE = 0, T_EMPTY = 1, T_WALL = 2, T_FLOOR = 3, T_DOOR = 4, T_KEY = 5, T_BALL = 6,
       T_BOX = 7, T_GOAL = 8, T_LAVA = 9, T_AGENT = 10 };
enum { C_RED = 0, C_GREEN = 1, C_BLUE = 2, C_PURPLE = 3, C_YELLOW = 4, C_GREY = 5 };
enum { A_LEFT = 0, A_RIGHT = 1, A_FORWARD = 2, A_PICKUP = 3, A_DROP = 4, A_TOGGLE = 5, A_DONE = 6 };

/* error bits, same values as include/mgrid.h */
enum { ERR_BAD_ACTION = 1, ERR_BAD_RESPAWN = 2, ERR_BAD_ORDER = 4, ERR_MISSING_DRAW = 8 };

/* One grid object.  type == T_NONE is Python's None.  For agents color = agent_id and
 * state = dir (Agent.encode, envs/gym_multigrid/multigrid.py:137-139); for doors state is
 * 0 open / 1 closed / 2 locked (multigrid.py:93). */
typedef struct { uint8_t type, color, state, pad; } obj_t;

static const obj_t NONE = {T_NONE, 0, 0, 0};

/* gym_minigrid.minigrid.DIR_TO_VEC (consistent with get_view_exts,
 * multigrid.py:754-769) */
/* WorldObj.encode()[0]; LavaWall (lava_walls.py:25-38) is a grey 'lava' object whose encoding is a grey wall's */
#define enc_type(o) (((o).type == T_LAVA && (o).color == C_GREY) ? T_WALL : (o).type)
static const int DIR_TO_VEC[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};

typedef struct { obj_t c[MAXW * MAXW]; int w, h; } grid_t;

/* minigrid Grid.get / Grid.set: storage index j*width + i */
static obj_t grid_get(const grid_t* g, int i, int j) { return g->c[j * g->w + i]; }
static void grid_set(grid_t* g, int i, int j, obj_t v) { g->c[j * g->w + i] = v; }

/* WorldObj flags */
static int can_overlap(obj_t o) {
  return o.type == T_GOAL || o.type == T_LAVA || o.type == T_FLOOR || (o.type == T_DOOR && o.state == 0);
}
static int can_pickup(obj_t o) { return o.type == T_KEY || o.type == T_BALL || o.type == T_BOX; }
static int see_behind(obj_t o) {
  if (o.type == T_WALL) return 0;
  if (o.type == T_DOOR) return o.state == 0;
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11) — the build's own free-running RNG; it has no
 * reference counterpart (the reference uses NumPy's MT19937 / PCG64, which replay mode
 * injects instead).  Counter layout: (env_global_id lo, step index lo, step index hi,
 * purpose<<28 | agent<<20 | attempt).
 * ---------------------------------------------------------------------------------------- */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void mgo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

enum { P_ORDER = 0, P_RESPAWN = 1, P_SPAWN = 2, P_ACTION = 3, P_GEN = 4, P_KEY = 5, P_OBJDROP = 6 };
enum { KIND_STATIC = 0, KIND_DOORKEY = 1, KIND_CLUTTERED = 2, KIND_MEETUP = 3, KIND_FOURROOMS = 4 };
#define MAXOBJ 64              /* objects a scatter generator places (goals/doors + clutter) */
#define MAXGOALS 6
#define MAX_ATTEMPTS 256

static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

typedef struct mgo_env {
  int W, H, A, V, max_steps, see_through, competitive, shaping, auto_reset;
  uint64_t seed, gid;
  grid_t base;                 /* static layout, restored by every reset (fixed_environment) */
  grid_t grid;
  int ax[MAXA], ay[MAXA], adir[MAXA], has_pos[MAXA];
  obj_t carrying[MAXA];
  int done[MAXA];
  int step_count;
  /* v8 shaper state (v8_robust_ppo.py:127-129) */
  int sh_has[MAXA]; double sh_dist[MAXA]; int sh_px[MAXA], sh_py[MAXA], sh_stat[MAXA];
  uint32_t err;
  /* draw source for place_one_agent during the current call */
  const int8_t* inj;           /* injected (x,y,dir) per agent, stride 4 bytes, or NULL */
  int purpose; uint64_t t;
  /* recorder: what the last step drew, in the layout replay mode injects */
  uint8_t last_order[MAXA];
  int8_t last_respawn[MAXA * 4], last_spawn[MAXA * 4];
  /* procedurally generated kinds (DoorKey): */
  int kind;                    /* KIND_* */
  int spawn_w;                 /* agents are (re)placed at x < spawn_w (doorkey.py:69-86); W otherwise */
  int door_y;                  /* DoorKey: door_idx of the last _gen_grid (doorkey.py:58-61), mirrored in the state word */
  const int8_t* inj_gen;       /* injected (split_idx, door_idx, key_x, key_y) or NULL */
  const int8_t* inj_obj;       /* injected (x, y, 0, 0) per agent for a carried object re-placed on death, or NULL */
  int8_t last_gen[4], last_objdrop[MAXA * 4];
  /* scatter generators (ClutteredMultiGrid with random layouts, MeetupEnv): */
  int n_goals;                 /* Cluttered: 1 = randomize_goal, 0 = goal at (W-2,H-2); Meetup: number of goal doors */
  int n_clutter, lava;         /* obstacles per layout; Cluttered walls_are_lava */
  int two_rooms;               /* FourRoomsEnv(two_rooms=True), fourrooms.py:32,75-85 */
  int full_obs;                /* fully_observed (multigrid.py:1043-1045, 1064-1065): Meetup */
  int goal_x[MAXGOALS], goal_y[MAXGOALS];   /* MeetupEnv.goal_pos (meetup.py:67-70) */
  int past_dist[MAXA];         /* MeetupEnv.past_goal_dist (meetup.py:61, 94) */
  const int8_t* inj_scatter;   /* injected accepted (x, y) of every scattered object, in draw order, or NULL */
  /* fixed start cells: mode 1 = EmptyEnv agent_start='fixed' (empty.py:33-36,72-89): every placement puts the agent
   * on its start cell facing east, whoever stands there goes back to its own start first; mode 2 = MazeEnv
   * (maze.py:86): reset places the agent on its start cell with a drawn direction, respawns stay random */
  int fixed_mode; int fixed_x[MAXA], fixed_y[MAXA];
  int8_t last_scatter[2 * MAXOBJ]; int n_scatter;
} mgo_env;

mgo_env* mgo_create(int W, int H, int A, int V, int max_steps, int see_through, int competitive,
                    int shaping, int auto_reset, uint64_t seed, uint64_t gid) {
  if (W < 3 || H < 3 || W > MAXW || H > MAXW || A < 1 || A > MAXA || V < 3 || V > MAXV || !(V & 1)) return NULL;
  mgo_env* e = (mgo_env*)calloc(1, sizeof(mgo_env));
  e->W = W; e->H = H; e->A = A; e->V = V; e->max_steps = max_steps; e->see_through = see_through;
  e->competitive = (A == 1) ? 1 : competitive;      /* multigrid.py:380-381 */
  e->shaping = shaping; e->auto_reset = auto_reset; e->seed = seed; e->gid = gid;
  e->base.w = e->grid.w = W; e->base.h = e->grid.h = H;
  e->kind = KIND_STATIC; e->spawn_w = W; e->door_y = 0;
  return e;
}
void mgo_set_kind(mgo_env* e, int kind) { e->kind = kind; e->full_obs = (kind == KIND_MEETUP); }
/* generator parameters of the scatter kinds; returns 0 if they do not fit */
int mgo_set_scatter(mgo_env* e, int n_goals, int n_clutter, int lava) {
  if (n_goals < 0 || n_goals > MAXGOALS || n_clutter < 0 || n_goals + n_clutter > MAXOBJ) return 0;
  if (e->kind == KIND_MEETUP && n_goals < 1) return 0;
  e->n_goals = n_goals; e->n_clutter = n_clutter; e->lava = lava;
  return 1;
}
void mgo_set_two_rooms(mgo_env* e, int on) { e->two_rooms = on; }
int mgo_obs_bytes(const mgo_env* e) { return e->full_obs ? e->W * e->H * 3 : e->V * e->V * 3; }
void mgo_destroy(mgo_env* e) { free(e); }
/* xy: [A][2] start cells; mode 0 switches the feature off */
void mgo_set_fixed_spawn(mgo_env* e, const int8_t* xy, int mode) {
  e->fixed_mode = mode;
  for (int a = 0; a < e->A && xy; ++a) { e->fixed_x[a] = xy[2 * a]; e->fixed_y[a] = xy[2 * a + 1]; }
}

/* cells: uint8[W*H*3], [(y*W+x)*3+c]; type 1 (empty) and 0 mean None */
void mgo_set_map(mgo_env* e, const uint8_t* cells) {
  for (int y = 0; y < e->H; ++y)
    for (int x = 0; x < e->W; ++x) {
      const uint8_t* p = cells + (y * e->W + x) * 3;
      obj_t o = {p[0], p[1], p[2], 0};
      if (o.type == T_EMPTY || o.type == T_NONE) o = NONE;
      grid_set(&e->base, x, y, o);
    }
}

/* ---- placement: place_obj / place_one_agent / place_agent_at_pos (multigrid.py:560-675) ---- */

static void draw_words(const mgo_env* e, int purpose, int agent, int attempt, uint32_t r[4]) {
  uint32_t ctr[4] = {(uint32_t)e->gid, (uint32_t)e->t, (uint32_t)(e->t >> 32),
                     ((uint32_t)purpose << 28) | ((uint32_t)agent << 20) | (uint32_t)attempt};
  uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
  philox4x32_10(ctr, key, r);
}

/* one candidate of place_obj's loop: x in [0, min(size_w, W)), y in [0, H) (multigrid.py:598-601;
 * top = (0,0), size = (xmax, height) — doorkey.py:56,64-65,77-78) */
static void draw_candidate_in(mgo_env* e, int purpose, int agent, int attempt, int xmax, int* x, int* y, int* dir) {
  uint32_t r[4];
  draw_words(e, purpose, agent, attempt, r);
  *x = (int)mulhi32(r[0], (uint32_t)(xmax < e->W ? xmax : e->W));
  *y = (int)mulhi32(r[1], (uint32_t)e->H);
  *dir = (int)(r[2] >> 30);
}

static void draw_candidate(mgo_env* e, int agent, int attempt, int* x, int* y, int* dir) {
  uint32_t r[4];
  draw_words(e, e->purpose, agent, attempt, r);
  *x = (int)mulhi32(r[0], (uint32_t)(e->spawn_w < e->W ? e->spawn_w : e->W));   /* _rand_int(0, min(size_w, grid.width)) multigrid.py:598-599 */
  *y = (int)mulhi32(r[1], (uint32_t)e->H);   /* _rand_int(0, grid.height) multigrid.py:600-601 */
  *dir = (int)(r[2] >> 30);                  /* _rand_int(0, 4)           multigrid.py:666 */
}

/* place_obj's acceptance test (multigrid.py:603-613): cell must hold nothing, and must not be
 * any agent's recorded position (agents being placed have agent_pos None, multigrid.py:655). */
static int cell_free(const mgo_env* e, int x, int y) {
  if (x < 0 || y < 0 || x >= e->W || y >= e->H) return 0;
  if (grid_get(&e->grid, x, y).type != T_NONE) return 0;
  for (int a = 0; a < e->A; ++a)
    if (e->has_pos[a] && e->ax[a] == x && e->ay[a] == y) return 0;
  return 1;
}

/* place_one_agent (multigrid.py:646-661) + place_agent_at_pos (multigrid.py:663-675) */
static void place_one_agent(mgo_env* e, int a) {
  if (e->fixed_mode == 1 || (e->fixed_mode == 2 && e->purpose == P_SPAWN)) {
    const int sx = e->fixed_x[a], sy = e->fixed_y[a];
    int dir = 0;                                       /* agent_start_dir = [0] * n_agents, empty.py:35 */
    if (e->fixed_mode == 1) {
      obj_t other = grid_get(&e->grid, sx, sy);        /* empty.py:82-85: move whoever is on my start cell back to its own */
      if (other.type == T_AGENT && other.color != a) place_one_agent(e, other.color);
    } else {
      int cx, cy;
      draw_candidate(e, a, 0, &cx, &cy, &dir);         /* place_agent_at_pos(0, start_pos): only the direction is drawn */
    }
    int8_t* rec = (e->purpose == P_SPAWN ? e->last_spawn : e->last_respawn) + 4 * a;
    rec[0] = (int8_t)sx; rec[1] = (int8_t)sy; rec[2] = (int8_t)dir; rec[3] = 0;
    e->ax[a] = sx; e->ay[a] = sy; e->has_pos[a] = 1; e->adir[a] = dir;
    obj_t ag = {T_AGENT, (uint8_t)a, (uint8_t)dir, 0};
    grid_set(&e->grid, sx, sy, ag);
    return;
  }
  e->has_pos[a] = 0;                                   /* multigrid.py:655 */
  int x = -1, y = -1, dir = 0;
  if (e->inj) {                                        /* replay: the accepted draw is injected */
    const int8_t* p = e->inj + 4 * a;
    x = p[0]; y = p[1]; dir = p[2];
    if (x < 0) { e->err |= ERR_MISSING_DRAW; x = -1; }
    else if (!cell_free(e, x, y) || x >= e->spawn_w || dir < 0 || dir > 3) { e->err |= ERR_BAD_RESPAWN; x = -1; }
  } else {
    for (int k = 0; k < MAX_ATTEMPTS; ++k) {           /* while True: multigrid.py:589-619 */
      int cx, cy, cd;
      draw_candidate(e, a, k, &cx, &cy, &cd);
      if (cell_free(e, cx, cy)) { x = cx; y = cy; dir = cd; break; }
    }
  }
  if (x < 0) {
    /* build-defined fallback so that the GPU loop has an exit condition: first free cell in
     * storage order, direction 0.  (The reference would loop forever, multigrid.py:644.) */
    dir = 0;
    for (int j = 0; j < e->H && x < 0; ++j)
      for (int i = 0; i < e->W && i < e->spawn_w; ++i)
        if (cell_free(e, i, j)) { x = i; y = j; break; }
    if (x < 0) { x = 0; y = 0; }
  }
  {
    int8_t* rec = (e->purpose == P_SPAWN ? e->last_spawn : e->last_respawn) + 4 * a;
    rec[0] = (int8_t)x; rec[1] = (int8_t)y; rec[2] = (int8_t)dir; rec[3] = 0;
  }
  e->ax[a] = x; e->ay[a] = y; e->has_pos[a] = 1;       /* multigrid.py:664 */
  e->adir[a] = dir;                                    /* multigrid.py:665-666 */
  obj_t ag = {T_AGENT, (uint8_t)a, (uint8_t)dir, 0};   /* multigrid.py:669-675 */
  grid_set(&e->grid, x, y, ag);
}

/* place_obj for a non-agent object (multigrid.py:560-627) inside x < xmax: injected position or
 * Philox rejection sampling with the given purpose; same bounded-loop fallback as agents. */
static void place_object(mgo_env* e, obj_t obj, int purpose, int agent, int xmax, const int8_t* inj, int8_t* rec) {
  int x = -1, y = -1, d;
  if (inj) {
    x = inj[0]; y = inj[1];
    if (x < 0) { e->err |= ERR_MISSING_DRAW; x = -1; }
    else if (!cell_free(e, x, y) || x >= xmax) { e->err |= ERR_BAD_RESPAWN; x = -1; }
  } else {
    for (int k = 0; k < MAX_ATTEMPTS; ++k) {
      int cx, cy;
      draw_candidate_in(e, purpose, agent, k, xmax, &cx, &cy, &d);
      if (cell_free(e, cx, cy)) { x = cx; y = cy; break; }
    }
  }
  if (x < 0) {
    for (int j = 0; j < e->H && x < 0; ++j)
      for (int i = 0; i < e->W && i < xmax; ++i)
        if (cell_free(e, i, j)) { x = i; y = j; break; }
    if (x < 0) { x = 0; y = 0; }
  }
  if (rec) { rec[0] = (int8_t)x; rec[1] = (int8_t)y; rec[2] = 0; rec[3] = 0; }
  grid_set(&e->grid, x, y, obj);
}

/* ---- observation: gen_obs_grid / slice / rotate_left / process_vis / encode ---- */

typedef struct { obj_t c[MAXV * MAXV]; int w, h; } view_t;
static obj_t view_get(const view_t* g, int i, int j) { return g->c[j * g->w + i]; }
static void view_set(view_t* g, int i, int j, obj_t v) { g->c[j * g->w + i] = v; }

/* Grid.slice (multigrid.py:301-319): out of bounds -> minigrid.Wall() (grey) */
static void grid_slice(const grid_t* g, int top_x, int top_y, int w, int h, view_t* out) {
  out->w = w; out->h = h;
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i) {
      int x = top_x + i, y = top_y + j;
      obj_t v;
      if (x >= 0 && x < g->w && y >= 0 && y < g->h) v = grid_get(g, x, y);
      else { v.type = T_WALL; v.color = C_GREY; v.state = 0; v.pad = 0; }
      view_set(out, i, j, v);
    }
}

/* Grid.rotate_left (multigrid.py:281-299) */
static void view_rotate_left(const view_t* in, view_t* out) {
  out->w = in->h; out->h = in->w;
  for (int i = 0; i < in->w; ++i)
    for (int j = 0; j < in->h; ++j) {
      obj_t v = view_get(in, i, j);
      if (v.type == T_AGENT) {                 /* multigrid.py:290-295 */
        int d = (int)v.state - 1;
        if (d < 0) d += 4;
        v.state = (uint8_t)d;
      }
      view_set(out, j, out->h - 1 - i, v);      /* multigrid.py:297 */
    }
}

/* minigrid Grid.process_vis — literal two-pass loops */
static void process_vis(view_t* g, int apx, int apy, uint8_t* mask /* [w*h], index i*h+j */) {
  int w = g->w, h = g->h;
  memset(mask, 0, (size_t)(w * h));
#define M(i, j) mask[(i) * h + (j)]
  M(apx, apy) = 1;
  for (int j = h - 1; j >= 0; --j) {
    for (int i = 0; i < w - 1; ++i) {
      if (!M(i, j)) continue;
      obj_t cell = view_get(g, i, j);
      if (cell.type != T_NONE && !see_behind(cell)) continue;
      M(i + 1, j) = 1;
      if (j > 0) { M(i + 1, j - 1) = 1; M(i, j - 1) = 1; }
    }
    for (int i = w - 1; i >= 1; --i) {
      if (!M(i, j)) continue;
      obj_t cell = view_get(g, i, j);
      if (cell.type != T_NONE && !see_behind(cell)) continue;
      M(i - 1, j) = 1;
      if (j > 0) { M(i - 1, j - 1) = 1; M(i, j - 1) = 1; }
    }
  }
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i)
      if (!M(i, j)) view_set(g, i, j, NONE);
#undef M
}
/* exported for the exhaustive bitmask-vs-literal test: opaque[i*V+j] != 0 -> wall */
void mgo_process_vis(int V, const uint8_t* opaque, uint8_t* mask) {
  view_t g; g.w = g.h = V;
  for (int i = 0; i < V; ++i)
    for (int j = 0; j < V; ++j) {
      obj_t o = NONE;
      if (opaque[i * V + j]) { o.type = T_WALL; o.color = C_GREY; }
      view_set(&g, i, j, o);
    }
  process_vis(&g, V / 2, V - 1, mask);
}

/* batched form over bit patterns: bit (i*V+j) of a pattern = cell (i,j) opaque; same layout out */
void mgo_process_vis_bits(int V, const uint64_t* patterns, int64_t n, uint64_t* out) {
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < n; ++k) {
    uint8_t op[MAXV * MAXV], m[MAXV * MAXV];
    for (int c = 0; c < V * V; ++c) op[c] = (uint8_t)((patterns[k] >> c) & 1u);
    mgo_process_vis(V, op, m);
    uint64_t r = 0;
    for (int c = 0; c < V * V; ++c) r |= (uint64_t)(m[c] != 0) << c;
    out[k] = r;
  }
}

/* gen_obs_grid (multigrid.py:997-1035) + gen_agent_obs (multigrid.py:1069-1083) +
 * minigrid Grid.encode.  image: uint8[V*V*3] indexed [(i*V+j)*3+c]. */
static void gen_agent_obs(const mgo_env* e, int a, uint8_t* image, uint8_t* direction) {
  int V = e->V, top_x, top_y;
  /* get_view_exts (multigrid.py:743-776) */
  switch (e->adir[a]) {
    case 0: top_x = e->ax[a];             top_y = e->ay[a] - V / 2;     break;
    case 1: top_x = e->ax[a] - V / 2;     top_y = e->ay[a];             break;
    case 2: top_x = e->ax[a] - V + 1;     top_y = e->ay[a] - V / 2;     break;
    default: top_x = e->ax[a] - V / 2;    top_y = e->ay[a] - V + 1;     break;
  }
  view_t g0, g1;
  grid_slice(&e->grid, top_x, top_y, V, V, &g0);                 /* multigrid.py:1012-1013 */
  view_t* cur = &g0; view_t* nxt = &g1;
  for (int k = 0; k < e->adir[a] + 1; ++k) {                     /* multigrid.py:1015-1016 */
    view_rotate_left(cur, nxt);
    view_t* t = cur; cur = nxt; nxt = t;
  }
  uint8_t mask[MAXV * MAXV];
  if (!e->see_through) process_vis(cur, V / 2, V - 1, mask);     /* multigrid.py:1020-1022 */
  else memset(mask, 1, (size_t)(V * V));                         /* multigrid.py:1024 */
  /* multigrid.py:1029-1033: own cell <- carried object or None, after process_vis */
  view_set(cur, V / 2, V - 1, e->carrying[a]);
  /* Grid.encode(vis_mask) */
  for (int i = 0; i < V; ++i)
    for (int j = 0; j < V; ++j) {
      uint8_t* p = image + (i * V + j) * 3;
      if (!mask[i * V + j]) { p[0] = p[1] = p[2] = 0; continue; }   /* unseen */
      obj_t v = view_get(cur, i, j);
      if (v.type == T_NONE) { p[0] = T_EMPTY; p[1] = 0; p[2] = 0; }
      else { p[0] = enc_type(v); p[1] = v.color; p[2] = v.state; }
    }
  *direction = (uint8_t)e->adir[a];                               /* multigrid.py:1083 */
}

/* ---- reset (multigrid.py:466-501; cluttered.py:25-40, 55-63) ---- */

static void clear_shaper(mgo_env* e) {                /* v8_robust_ppo.py:227-229 */
  for (int a = 0; a < e->A; ++a) { e->sh_has[a] = 0; e->sh_dist[a] = 0.0; e->sh_px[a] = e->sh_py[a] = 0; e->sh_stat[a] = 0; }
}

/* DoorKeyEnv._gen_grid (multigrid_envs/doorkey.py:34-67).  Draw order of the reference: split_idx,
 * agents (x < split_idx), door_idx, key (x < split_idx). */
static void gen_doorkey(mgo_env* e) {
  const int W = e->W, H = e->H;
  const obj_t wall = {T_WALL, C_GREY, 0, 0}, goal = {T_GOAL, C_GREEN, 0, 0};
  for (int j = 0; j < H; ++j)
    for (int i = 0; i < W; ++i)                          /* Grid(width,height); wall_rect(0,0,w,h) :38-41 */
      grid_set(&e->grid, i, j, (i == 0 || j == 0 || i == W - 1 || j == H - 1) ? wall : NONE);
  grid_set(&e->grid, W - 2, H - 2, goal);                /* put_obj(Goal, w-2, h-2) :44 */
  int split, door;
  if (e->inj_gen) { split = e->inj_gen[0]; door = e->inj_gen[1]; }
  else {
    uint32_t r[4];
    draw_words(e, P_GEN, 0, 0, r);
    const int start = W <= 5 ? 2 : 3;                    /* :47-50 */
    split = start + (int)mulhi32(r[0], (uint32_t)(W - 2 - start));   /* _rand_int(start_idx, width-2) :51 */
    door = 1 + (int)mulhi32(r[1], (uint32_t)(W - 2 - 1));            /* _rand_int(1, width-2) :59 */
  }
  if (split < 1 || split > W - 2 || door < 1 || door > H - 2) { e->err |= ERR_BAD_RESPAWN; split = W <= 5 ? 2 : 3; door = 1; }
  for (int j = 0; j < H; ++j) grid_set(&e->grid, split, j, wall);    /* vert_wall(split_idx, 0) :52 */
  e->spawn_w = split;
  e->door_y = door;
  for (int a = 0; a < e->A; ++a) place_one_agent(e, a);  /* place_agent(size=(split_idx,height)) :56 */
  const obj_t doorobj = {T_DOOR, C_YELLOW, 2, 0}, keyobj = {T_KEY, C_YELLOW, 0, 0};
  grid_set(&e->grid, split, door, doorobj);              /* put_obj(Door('yellow', is_locked=True)) :60-61 */
  place_object(e, keyobj, P_KEY, 0, split, e->inj_gen ? e->inj_gen + 2 : NULL, e->last_gen + 2);   /* :64-65 */
  e->last_gen[0] = (int8_t)split; e->last_gen[1] = (int8_t)door;
}

/* place_obj(obj, max_tries=100) anywhere on the grid for scattered object number idx (cluttered.py:28-36,
 * meetup.py:67-72).  The reference raises after 100 failed tries; here, as everywhere, the loop is
 * bounded at MAX_ATTEMPTS and then falls back to the first free cell (unreachable in practice:
 * more than half of every such grid is free). */
static void place_scatter(mgo_env* e, obj_t obj, int idx, int* ox, int* oy) {
  int8_t rec[4], inj[4] = {0, 0, 0, 0};
  if (e->inj_scatter) { inj[0] = e->inj_scatter[2 * idx]; inj[1] = e->inj_scatter[2 * idx + 1]; }
  place_object(e, obj, P_GEN, idx, e->W, e->inj_scatter ? inj : NULL, rec);
  e->last_scatter[2 * idx] = rec[0]; e->last_scatter[2 * idx + 1] = rec[1];
  if (ox) { *ox = rec[0]; *oy = rec[1]; }
}

/* MeetupEnv.get_dist (meetup.py:78-84): Manhattan distance of every agent to the goal whose summed
 * distance over the agents is smallest (np.argmin: the first such goal) */
static void meetup_dist(const mgo_env* e, int* out) {
  int best = 0, best_sum = 0;
  for (int i = 0; i < e->n_goals; ++i) {
    int sum = 0;
    for (int j = 0; j < e->A; ++j) sum += abs(e->goal_x[i] - e->ax[j]) + abs(e->goal_y[i] - e->ay[j]);
    if (i == 0 || sum < best_sum) { best = i; best_sum = sum; }
  }
  for (int j = 0; j < e->A; ++j) out[j] = abs(e->goal_x[best] - e->ax[j]) + abs(e->goal_y[best] - e->ay[j]);
}

/* ClutteredMultiGrid._gen_grid (multigrid_envs/cluttered.py:25-40) and MeetupEnv._gen_grid
 * (multigrid_envs/meetup.py:64-76): border walls, then objects scattered by place_obj, then agents. */
static void gen_scatter(mgo_env* e) {
  const int W = e->W, H = e->H;
  const obj_t wall = {T_WALL, C_GREY, 0, 0}, goal = {T_GOAL, C_GREEN, 0, 0}, lava = {T_LAVA, C_RED, 0, 0};
  const obj_t door = {T_DOOR, C_RED, 2, 0};                       /* Door(color='red', is_locked=True) meetup.py:69 */
  for (int j = 0; j < H; ++j)
    for (int i = 0; i < W; ++i)
      grid_set(&e->grid, i, j, (i == 0 || j == 0 || i == W - 1 || j == H - 1) ? wall : NONE);
  int idx = 0;
  if (e->kind == KIND_CLUTTERED) {
    if (e->n_goals) place_scatter(e, goal, idx++, NULL, NULL);    /* cluttered.py:28-29 */
    else grid_set(&e->grid, W - 2, H - 2, goal);                  /* cluttered.py:31 */
    const obj_t lavawall = {T_LAVA, C_GREY, 0, 0};                /* lava_walls.py:79-80 */
    for (int c = 0; c < e->n_clutter; ++c) place_scatter(e, e->lava == 2 ? lavawall : e->lava ? lava : wall, idx++, NULL, NULL);   /* :32-36 */
  } else {
    for (int i = 0; i < e->n_goals; ++i) place_scatter(e, door, idx++, &e->goal_x[i], &e->goal_y[i]);    /* meetup.py:67-70 */
    for (int c = 0; c < e->n_clutter; ++c) place_scatter(e, wall, idx++, NULL, NULL);                   /* :71-72 */
  }
  e->n_scatter = idx;
  e->spawn_w = W;
  for (int a = 0; a < e->A; ++a) place_one_agent(e, a);           /* place_agent() cluttered.py:38 / meetup.py:74 */
  if (e->kind == KIND_MEETUP) meetup_dist(e, e->past_dist);       /* MeetupEnv.reset meetup.py:59-62 */
}

/* FourRoomsEnv._gen_grid with two_rooms=False and random agents / goal (multigrid_envs/fourrooms.py:53-104):
 * border walls, a vertical wall at x = W//2 and a horizontal wall at y = H//2, one gap per wall segment
 * (drawn in the order top, left, right, bottom; a gap at x <= 1, x >= W-1, y <= 0 or y >= H-1 is not carved,
 * :79-80,88-89), then the agents, then the goal by place_obj.  Recorded / injected as three (a, b) pairs:
 * (gap_top_y, gap_left_x), (gap_right_x, gap_bottom_y), (goal_x, goal_y). */
static void gen_fourrooms(mgo_env* e) {
  const int W = e->W, H = e->H, rw = W / 2, rh = H / 2;
  const obj_t wall = {T_WALL, C_GREY, 0, 0}, goal = {T_GOAL, C_GREEN, 0, 0};
  for (int j = 0; j < H; ++j)
    for (int i = 0; i < W; ++i)                          /* horz_wall / vert_wall around the grid :58-61 */
      grid_set(&e->grid, i, j, (i == 0 || j == 0 || i == W - 1 || j == H - 1) ? wall : NONE);
  int g[4];                                              /* top y, left x, right x, bottom y */
  if (e->inj_scatter) { g[0] = e->inj_scatter[0]; g[1] = e->inj_scatter[1]; g[2] = e->inj_scatter[2]; g[3] = e->inj_scatter[3]; }
  else {
    uint32_t r[4];
    draw_words(e, P_GEN, 0, 0, r);
    g[0] = 1 + (int)mulhi32(r[0], (uint32_t)(rh - 1));           /* _rand_int(y_top + 1, y_bottom), j = 0   :77 */
    g[1] = 1 + (int)mulhi32(r[1], (uint32_t)(rw - 1));           /* _rand_int(x_left + 1, x_right), i = 0   :86 */
    g[2] = rw + 1 + (int)mulhi32(r[2], (uint32_t)(rw - 1));      /* i = 1 */
    g[3] = rh + 1 + (int)mulhi32(r[3], (uint32_t)(rh - 1));      /* j = 1 */
  }
  /* two_rooms (fourrooms.py:75-83): only the vertical wall; `not (j == 1 and two_rooms and height < 7)` skips the lower
   * gap's draw on small grids.  The unused entries take fixed in-range values so that the records stay comparable. */
  const int two = e->two_rooms, low = !(two && H < 7);
  if (two) { g[1] = 1; g[2] = rw + 1; }
  if (!low) g[3] = rh + 1;
  if (g[0] < 1 || g[0] >= rh || g[1] < 1 || g[1] >= rw || g[2] <= rw || g[2] >= 2 * rw || g[3] <= rh || g[3] >= 2 * rh) {
    e->err |= ERR_BAD_RESPAWN; g[0] = 1; g[1] = 1; g[2] = rw + 1; g[3] = rh + 1;
  }
  const int gx[4] = {rw, g[1], g[2], rw}, gy[4] = {g[0], rh, rh, g[3]};
  /* the loop of :68-90, in its order: (j=0,i=0) vertical top + gap, horizontal left + gap; (j=0,i=1) horizontal
   * right + gap; (j=1,i=0) vertical bottom + gap */
  for (int step = 0; step < 4; ++step) {
    if (two && (step == 1 || step == 2)) continue;                                             /* `if not self.two_rooms` :85 */
    if (step == 0) for (int y = 0; y < rh; ++y) grid_set(&e->grid, rw, y, wall);            /* vert_wall(x_right, y_top, room_h) */
    if (step == 1) for (int x = 0; x < rw; ++x) grid_set(&e->grid, x, rh, wall);            /* horz_wall(x_left, y_bottom, room_w) */
    if (step == 2) for (int x = rw; x < 2 * rw; ++x) grid_set(&e->grid, x, rh, wall);
    if (step == 3) for (int y = rh; y < 2 * rh; ++y) grid_set(&e->grid, rw, y, wall);
    if (step == 3 && !low) continue;                                                           /* :76 */
    if (!(gx[step] <= 1 || gx[step] >= W - 1 || gy[step] <= 0 || gy[step] >= H - 1)) grid_set(&e->grid, gx[step], gy[step], NONE);
  }
  e->last_scatter[0] = (int8_t)g[0]; e->last_scatter[1] = (int8_t)g[1];
  e->last_scatter[2] = (int8_t)g[2]; e->last_scatter[3] = (int8_t)g[3];
  e->spawn_w = W;
  for (int a = 0; a < e->A; ++a) place_one_agent(e, a);                                      /* place_agent() :98 */
  int8_t rec[4], inj[4] = {0, 0, 0, 0};
  if (e->inj_scatter) { inj[0] = e->inj_scatter[4]; inj[1] = e->inj_scatter[5]; }
  place_object(e, goal, P_GEN, 1, W, e->inj_scatter ? inj : NULL, rec);                      /* place_obj(Goal()) :104 */
  e->last_scatter[4] = rec[0]; e->last_scatter[5] = rec[1];
  e->n_scatter = 3;
}

static void do_reset(mgo_env* e, const int8_t* spawn, uint64_t t) {
  for (int a = 0; a < e->A; ++a) { e->has_pos[a] = 0; e->done[a] = 0; }   /* multigrid.py:471-473 */
  e->inj = spawn; e->purpose = P_SPAWN; e->t = t;
  if (e->kind == KIND_DOORKEY) gen_doorkey(e);
  else if (e->kind == KIND_CLUTTERED || e->kind == KIND_MEETUP) gen_scatter(e);
  else if (e->kind == KIND_FOURROOMS) gen_fourrooms(e);
  else {
    e->grid = e->base;                                   /* _gen_grid with the fixed seed */
    for (int a = 0; a < e->A; ++a) place_one_agent(e, a); /* cluttered.py:61-63 */
  }
  for (int a = 0; a < e->A; ++a) e->carrying[a] = NONE; /* multigrid.py:490 */
  e->step_count = 0;                                    /* multigrid.py:493 */
  clear_shaper(e);
}

void mgo_reset(mgo_env* e, const int8_t* spawn /* [A][4] or NULL */, uint64_t t) { e->inj_gen = NULL; do_reset(e, spawn, t); }
void mgo_reset_gen(mgo_env* e, const int8_t* spawn, const int8_t* gen /* [4] or NULL */, uint64_t t) {
  e->inj_gen = gen; do_reset(e, spawn, t); e->inj_gen = NULL;
}

void mgo_reset_scatter(mgo_env* e, const int8_t* spawn, const int8_t* scatter /* [n_obj][2] or NULL */, uint64_t t) {
  e->inj_gen = NULL; e->inj_scatter = scatter; do_reset(e, spawn, t); e->inj_scatter = NULL;
}

/* image: [A][obs_bytes].  fully_observed (multigrid.py:1043-1045): every agent gets Grid.encode() of
 * the whole grid, [i][j][c] with the Agent objects in their cells, and its own direction. */
void mgo_get_obs(const mgo_env* e, uint8_t* image, uint8_t* dir /* [A] */) {
  if (e->full_obs) {
    const int cb = e->W * e->H * 3;
    for (int a = 0; a < e->A; ++a) {
      for (int i = 0; i < e->W; ++i)
        for (int j = 0; j < e->H; ++j) {
          obj_t v = grid_get(&e->grid, i, j);
          uint8_t* p = image + a * cb + (i * e->H + j) * 3;
          if (v.type == T_NONE) { p[0] = T_EMPTY; p[1] = 0; p[2] = 0; } else { p[0] = enc_type(v); p[1] = v.color; p[2] = v.state; }
        }
      dir[a] = (uint8_t)e->adir[a];
    }
    return;
  }
  for (int a = 0; a < e->A; ++a) gen_agent_obs(e, a, image + a * e->V * e->V * 3, dir + a);
}

/* ---- step (multigrid.py:815-995) ---- */

/* agent_is_done (multigrid.py:815-832) */
static void agent_is_done(mgo_env* e, int a) {
  grid_set(&e->grid, e->ax[a], e->ay[a], NONE);         /* multigrid.py:817-819 */
  e->done[a] = 1;                                       /* multigrid.py:821 */
  e->last_objdrop[4 * a] = -1; e->last_objdrop[4 * a + 1] = -1;
  if (e->carrying[a].type != T_NONE) {
    /* multigrid.py:827-829: place_obj(carrying) anywhere on the grid, before the agent respawns */
    place_object(e, e->carrying[a], P_OBJDROP, a, e->W, e->inj_obj ? e->inj_obj + 4 * a : NULL, e->last_objdrop + 4 * a);
    e->carrying[a] = NONE;
  }
  place_one_agent(e, a);                                /* multigrid.py:832 */
}

/* move_agent (multigrid.py:834-847) */
static void move_agent(mgo_env* e, int a, int nx, int ny) {
  obj_t ag = grid_get(&e->grid, e->ax[a], e->ay[a]);
  grid_set(&e->grid, e->ax[a], e->ay[a], NONE);
  e->ax[a] = nx; e->ay[a] = ny;
  grid_set(&e->grid, nx, ny, ag);                       /* overwrites whatever was there */
}

/* rotate_agent (multigrid.py:849-858) */
static void rotate_agent(mgo_env* e, int a) {
  obj_t ag = grid_get(&e->grid, e->ax[a], e->ay[a]);
  ag.state = (uint8_t)e->adir[a];
  grid_set(&e->grid, e->ax[a], e->ay[a], ag);
}

/* _forward (multigrid.py:906-924) */
static int forward(mgo_env* e, int a, int fx, int fy) {
  obj_t fwd = grid_get(&e->grid, fx, fy);
  int blocking = 0;
  for (int b = 0; b < e->A; ++b)
    if (b != a && e->ax[b] == fx && e->ay[b] == fy) blocking = 1;
  if (!blocking) {
    if (fwd.type == T_GOAL) agent_is_done(e, a);
    else if (fwd.type == T_LAVA) agent_is_done(e, a);
    else if (fwd.type == T_NONE || can_overlap(fwd)) move_agent(e, a, fx, fy);
    return 1;
  }
  return 0;
}

/* MiniGridEnv._reward */
static double env_reward(const mgo_env* e) {
  return 1 - 0.9 * ((double)e->step_count / (double)e->max_steps);
}
double mgo_reward_value(int step_count, int max_steps) { return 1 - 0.9 * ((double)step_count / (double)max_steps); }

/* step_one_agent (multigrid.py:860-904) */
static double step_one_agent(mgo_env* e, int action, int a) {
  double reward = 0;
  /* front_pos (multigrid.py:697-703), computed before the action */
  int fx = e->ax[a] + DIR_TO_VEC[e->adir[a]][0], fy = e->ay[a] + DIR_TO_VEC[e->adir[a]][1];
  /* agents live strictly inside the border wall, so (fx,fy) is always on the grid */
  if (action == A_LEFT) {
    e->adir[a] -= 1;
    if (e->adir[a] < 0) e->adir[a] += 4;
    rotate_agent(e, a);
  } else if (action == A_RIGHT) {
    e->adir[a] = (e->adir[a] + 1) % 4;
    rotate_agent(e, a);
  } else if (action == A_FORWARD) {
    int ok = forward(e, a, fx, fy);
    obj_t fwd = grid_get(&e->grid, fx, fy);             /* re-read after the move, :881 */
    if (ok && fwd.type == T_GOAL) reward = env_reward(e);
  } else if (action == A_PICKUP) {                      /* _pickup multigrid.py:926-938 */
    obj_t fwd = grid_get(&e->grid, fx, fy);
    if (fwd.type != T_NONE && can_pickup(fwd) && e->carrying[a].type == T_NONE) {
      e->carrying[a] = fwd;
      grid_set(&e->grid, fx, fy, NONE);
    }
  } else if (action == A_DROP) {                        /* _drop multigrid.py:940-951 */
    obj_t fwd = grid_get(&e->grid, fx, fy);
    if (fwd.type == T_NONE && e->carrying[a].type != T_NONE) {
      grid_set(&e->grid, fx, fy, e->carrying[a]);
      e->carrying[a] = NONE;
    }
  } else if (action == A_TOGGLE) {                      /* _toggle multigrid.py:953-961 */
    obj_t fwd = grid_get(&e->grid, fx, fy);
    if (fwd.type == T_DOOR) {                           /* Door.toggle multigrid.py:112-122 */
      if (fwd.state == 2) {
        if (e->carrying[a].type == T_KEY && e->carrying[a].color == fwd.color) { fwd.state = 0; grid_set(&e->grid, fx, fy, fwd); }
      } else {
        fwd.state = fwd.state == 0 ? 1 : 0;
        grid_set(&e->grid, fx, fy, fwd);
      }
    } else if (fwd.type == T_BOX) {
      grid_set(&e->grid, fx, fy, NONE);                 /* Box.toggle with empty contents */
    }
  } else if (action == A_DONE) {
    /* no-op, multigrid.py:898-899 */
  } else {
    e->err |= ERR_BAD_ACTION;                           /* assert False, multigrid.py:902 */
  }
  return reward;
}

/* get_goal_position (v8_robust_ppo.py:159-169) */
static void goal_position(const mgo_env* e, int* gx, int* gy) {
  for (int i = 0; i < e->W; ++i)
    for (int j = 0; j < e->H; ++j)
      if (grid_get(&e->grid, i, j).type == T_GOAL) { *gx = i; *gy = j; return; }
  *gx = 13; *gy = 13;
}
void mgo_goal(const mgo_env* e, int* gx, int* gy) { goal_position(e, gx, gy); }

/* shape_rewards (v8_robust_ppo.py:171-220), float64 with one rounding per Python operation */
static void shape_rewards(mgo_env* e, const double* orig, const int* actions, double* shaped) {
  int gx, gy;
  goal_position(e, &gx, &gy);
  for (int i = 0; i < e->A; ++i) {
    int px = e->ax[i], py = e->ay[i], action = actions[i];
    if (orig[i] > 0) {                                   /* v8:181-186 */
      shaped[i] = 5.0;
      e->sh_has[i] = 0; e->sh_stat[i] = 0;
      continue;
    }
    volatile double reward = 0.0;                        /* volatile: forbid FMA contraction */
    long dx = px - gx, dy = py - gy;
    double current_dist = sqrt((double)(dx * dx + dy * dy));   /* np.linalg.norm, v8:192 */
    if (e->sh_has[i]) {                                  /* v8:193-195 */
      volatile double dist_change = e->sh_dist[i] - current_dist;
      volatile double term = dist_change * 0.2;
      reward = reward + term;
    }
    int had_prev = e->sh_has[i];
    e->sh_dist[i] = current_dist;                        /* v8:196 */
    if (had_prev) {                                      /* v8:199-206 (prev_positions set iff prev_distances set) */
      if (px == e->sh_px[i] && py == e->sh_py[i]) {
        e->sh_stat[i] += 1;
        if (e->sh_stat[i] > 3) reward = reward - 0.05;
      } else {
        e->sh_stat[i] = 0;
        reward = reward + 0.02;
      }
    }
    if (action == 2) reward = reward + 0.02;             /* v8:209-212 */
    else if (action == 0 || action == 1) reward = reward + 0.01;
    double r = reward;
    if (-0.2 > r) r = -0.2;                              /* max(reward, -0.2), v8:215 */
    shaped[i] = r;
    e->sh_px[i] = px; e->sh_py[i] = py; e->sh_has[i] = 1; /* v8:218 */
  }
}

/* StableRewardShaper.shape_rewards (v7_stable_goal_focused_ppo.py:60-108), float64 with one rounding per
 * Python operation.  Unlike v8, a goal touch `continue`s past every prev_* update (v7:70-72), so the
 * shaper keeps what it held; the caller clears it at env.reset() only (v7:291,338). */
static void shape_rewards_v7(mgo_env* e, const double* orig, const int* actions, double* shaped) {
  int gx, gy;
  goal_position(e, &gx, &gy);                            /* v7:47-58, same search and (13,13) default as v8 */
  for (int i = 0; i < e->A; ++i) {
    int px = e->ax[i], py = e->ay[i], action = actions[i];
    if (orig[i] > 0) { shaped[i] = 10.0; continue; }     /* v7:70-72 */
    volatile double reward = 0.0;                        /* v7:75 */
    long dx = px - gx, dy = py - gy;
    double current_dist = sqrt((double)(dx * dx + dy * dy));   /* v7:78 */
    if (e->sh_has[i]) {                                  /* v7:79-81 */
      volatile double dist_change = e->sh_dist[i] - current_dist;
      volatile double term = dist_change * 0.2;
      reward = reward + term;
    }
    e->sh_dist[i] = current_dist;                        /* v7:82 */
    if (e->sh_has[i]) {                                  /* v7:85-93 (prev_positions set iff prev_distances set) */
      if (px == e->sh_px[i] && py == e->sh_py[i]) {
        e->sh_stat[i] += 1;
        if (e->sh_stat[i] > 3) {                         /* min(0.05, count * 0.01) */
          volatile double pen = (double)e->sh_stat[i] * 0.01;
          reward = reward - (pen < 0.05 ? pen : 0.05);
        }
      } else {
        e->sh_stat[i] = 0;
        reward = reward + 0.01;
      }
    }
    if (action == 2) reward = reward + 0.02;             /* v7:96-99 */
    else if (action == 0 || action == 1) reward = reward + 0.005;
    double r = reward;
    if (-0.2 > r) r = -0.2;                              /* max(shaped, -0.2), v7:104 */
    shaped[i] = r;
    e->sh_px[i] = px; e->sh_py[i] = py; e->sh_has[i] = 1; /* v7:106 */
  }
}

/* MultiGridEnv.step (multigrid.py:963-995) followed by the v8 shaper (v8:239-243) and the
 * build's auto-reset.  order == NULL -> Philox Fisher-Yates; respawn/reset_spawn == NULL ->
 * Philox rejection sampling.  t = step index (Philox counter).  Outputs may be NULL. */
void mgo_step_ex(mgo_env* e, const int* actions, const uint8_t* order, const int8_t* respawn,
                 const int8_t* objdrop, const int8_t* reset_spawn, const int8_t* reset_gen, uint64_t t,
                 uint8_t* image, uint8_t* dir, double* reward, double* env_rew, uint8_t* done_out);
void mgo_step(mgo_env* e, const int* actions, const uint8_t* order, const int8_t* respawn,
              const int8_t* reset_spawn, uint64_t t, uint8_t* image, uint8_t* dir, double* reward,
              double* env_rew, uint8_t* done_out) {
  mgo_step_ex(e, actions, order, respawn, NULL, reset_spawn, NULL, t, image, dir, reward, env_rew, done_out);
}
/* objdrop: injected (x,y,0,0) per agent for the carried object place_obj re-places when that agent
 * finishes (multigrid.py:827-829); reset_gen: injected DoorKey layout draws for an auto-reset. */
void mgo_step_ex(mgo_env* e, const int* actions, const uint8_t* order, const int8_t* respawn,
                 const int8_t* objdrop, const int8_t* reset_spawn, const int8_t* reset_gen, uint64_t t,
                 uint8_t* image, uint8_t* dir, double* reward, double* env_rew, uint8_t* done_out) {
  int A = e->A, ord[MAXA];
  e->inj_obj = objdrop;
  e->step_count += 1;                                    /* multigrid.py:968 */
  double rewards[MAXA];
  for (int a = 0; a < A; ++a) rewards[a] = 0;            /* multigrid.py:970 */
  if (order) {
    unsigned seen = 0;
    for (int k = 0; k < A; ++k) { ord[k] = order[k]; if (ord[k] < A) seen |= 1u << ord[k]; }
    if (seen != (1u << A) - 1) { e->err |= ERR_BAD_ORDER; for (int k = 0; k < A; ++k) ord[k] = k; }
  } else {
    /* uniform permutation; same shape as np.random.shuffle's Fisher-Yates (multigrid.py:973-974) */
    uint32_t ctr[4] = {(uint32_t)e->gid, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)P_ORDER << 28};
    uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)}, r[4];
    philox4x32_10(ctr, key, r);
    for (int k = 0; k < A; ++k) ord[k] = k;
    for (int i = A - 1, w = 0; i >= 1; --i, ++w) {
      int j = (int)mulhi32(r[w & 3], (uint32_t)(i + 1));
      int tmp = ord[i]; ord[i] = ord[j]; ord[j] = tmp;
    }
  }
  e->inj = respawn; e->purpose = P_RESPAWN; e->t = t;
  for (int a = 0; a < A; ++a) {
    e->last_order[a] = (uint8_t)ord[a];
    e->last_respawn[4 * a] = -1; e->last_respawn[4 * a + 1] = -1; e->last_respawn[4 * a + 2] = 0; e->last_respawn[4 * a + 3] = 0;
    e->last_objdrop[4 * a] = -1; e->last_objdrop[4 * a + 1] = -1; e->last_objdrop[4 * a + 2] = 0; e->last_objdrop[4 * a + 3] = 0;
  }
  for (int k = 0; k < A; ++k) {                          /* multigrid.py:977-978 */
    int a = ord[k];
    rewards[a] = step_one_agent(e, actions[a], a);
  }
  int collective_done = 0;                               /* multigrid.py:986-993 */
  if (e->competitive) {
    int s = 0;
    for (int a = 0; a < A; ++a) s += e->done[a];
    collective_done = s >= 1;
  }
  if (e->step_count >= e->max_steps) collective_done = 1;
  if (e->kind == KIND_MEETUP) {                          /* MeetupEnv.step (meetup.py:86-95) */
    int dist[MAXA], sum = 0;
    meetup_dist(e, dist);
    for (int a = 0; a < A; ++a) { rewards[a] = (double)(e->past_dist[a] - dist[a]); sum += dist[a]; }
    if (sum == A) {
      for (int a = 0; a < A; ++a) rewards[a] = rewards[a] + 1;
      collective_done = 1;
    }
    for (int a = 0; a < A; ++a) e->past_dist[a] = dist[a];
  }

  double shaped[MAXA];
  if (e->shaping == 1) shape_rewards(e, rewards, actions, shaped);
  else if (e->shaping == 2) shape_rewards_v7(e, rewards, actions, shaped);
  else for (int a = 0; a < A; ++a) shaped[a] = rewards[a];

  if (collective_done && e->auto_reset) { e->inj_gen = reset_gen; do_reset(e, reset_spawn, t); e->inj_gen = NULL; }
  e->inj_obj = NULL;

  if (image && dir) mgo_get_obs(e, image, dir);          /* multigrid.py:980 */
  for (int a = 0; a < A; ++a) {
    if (reward) reward[a] = shaped[a];
    if (env_rew) env_rew[a] = rewards[a];
  }
  if (done_out) *done_out = (uint8_t)collective_done;
}

/* Philox action draw for synthetic rollouts: uniform in [0,6].  One block per env-step: the actions are the base-7
 * digits of word 3 of the SAME block the agent order is drawn from (words 0..2, see mgo_step_ex) - digit a = multiply-high
 * by 7, carry on with the low half of the product.  Five agents use all four words for the order; their actions are the
 * five digits of word 0 of a second block (P_ACTION). */
void mgo_random_actions(const mgo_env* e, uint64_t t, int* actions) {
  uint32_t ctr[4] = {(uint32_t)e->gid, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)(e->A > 4 ? P_ACTION : P_ORDER) << 28};
  uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)}, r[4];
  philox4x32_10(ctr, key, r);
  uint32_t w = e->A > 4 ? r[0] : r[3];
  for (int a = 0; a < e->A; ++a) {
    actions[a] = (int)mulhi32(w, 7u);
    w *= 7u;
  }
}

/* state getters (same packing as include/mgrid.h mg_buffers.env_state): agent word =
 * x:6 | y:6 | dir:2 | prev_x:6 | prev_y:6 | valid:1 | stationary:3 (saturating at 7: the shaper
 * only ever tests "> 3", v8_robust_ppo.py:202, so the comparison view may saturate) */
uint32_t mgo_agent_state(const mgo_env* e, int a) {
  uint32_t w = (uint32_t)e->ax[a] | ((uint32_t)e->ay[a] << 6) | ((uint32_t)e->adir[a] << 12);
  if (e->sh_has[a]) {
    uint32_t st = (uint32_t)(e->sh_stat[a] > 7 ? 7 : e->sh_stat[a]);
    w |= ((uint32_t)e->sh_px[a] << 14) | ((uint32_t)e->sh_py[a] << 20) | (1u << 26) | (st << 27);
  }
  return w;
}
uint32_t mgo_shaper_state(const mgo_env* e, int a) {
  if (!e->sh_has[a]) return (uint32_t)e->sh_stat[a] << 17;
  return (uint32_t)e->sh_px[a] | ((uint32_t)e->sh_py[a] << 8) | (1u << 16) | ((uint32_t)e->sh_stat[a] << 17);
}
uint32_t mgo_step_state(const mgo_env* e) {
  uint32_t d = 0;
  for (int a = 0; a < e->A; ++a) d |= (uint32_t)(e->done[a] != 0) << a;
  uint32_t w = (uint32_t)e->step_count | (d << 16);
  /* DoorKey keeps the spawn region (split_idx) in bits 24..29 and the low five bits of the door's row (door_idx) in
   * bits 21..23 and 30..31 (include/mgrid.h) */
  if (e->kind == KIND_DOORKEY)
    w |= ((uint32_t)e->spawn_w << 24) | (((uint32_t)e->door_y & 7u) << 21) | ((((uint32_t)e->door_y >> 3) & 3u) << 30);
  return w;
}
/* One-byte cell kind used by the product's per-env grids: 0 = None, door = 128 + color*4 + state,
 * anything else = type*8 + color (state is 0 for every non-door object). */
static uint8_t kind_of(obj_t o) {
  if (o.type == T_NONE || o.type == T_AGENT) return 0;
  if (o.type == T_DOOR) return (uint8_t)(128 + o.color * 4 + o.state);
  return (uint8_t)(o.type * 8 + o.color);
}
/* all agents, byte a = kind carried by agent a (five agents need more than one word) */
uint64_t mgo_carry_word64(const mgo_env* e) {
  uint64_t w = 0;
  for (int a = 0; a < e->A && a < 8; ++a) w |= (uint64_t)kind_of(e->carrying[a]) << (8 * a);
  return w;
}
uint32_t mgo_carry_word(const mgo_env* e) {
  uint32_t w = 0;
  for (int a = 0; a < e->A && a < 4; ++a) w |= (uint32_t)kind_of(e->carrying[a]) << (8 * a);
  return w;
}
void mgo_grid_kinds(const mgo_env* e, uint8_t* out /* [H*W], index y*W+x; agents' cells read 0 */) {
  for (int y = 0; y < e->H; ++y)
    for (int x = 0; x < e->W; ++x) out[y * e->W + x] = kind_of(grid_get(&e->grid, x, y));
}
void mgo_last_gen(const mgo_env* e, int8_t* gen /* [4] */, int8_t* objdrop /* [A][4] */) {
  if (gen) memcpy(gen, e->last_gen, 4);
  if (objdrop) memcpy(objdrop, e->last_objdrop, (size_t)e->A * 4);
}
int mgo_last_scatter(const mgo_env* e, int8_t* out /* [MAXOBJ][2] */) {
  memcpy(out, e->last_scatter, (size_t)e->n_scatter * 2);
  return e->n_scatter;
}
/* kind-specific state in the product's packing (mg_buffers.aux, 4 words per env): words 0..2 = the goals, two per
 * word (goal 2k in the low half of word k, goal 2k+1 in the high half, each x | y<<6, 0xFFFF if unused), word 3 =
 * past_goal_dist, one byte per agent */
static void aux_put_goal(uint32_t* out, int i, int x, int y) {
  const uint32_t sh = 16u * (uint32_t)(i & 1);
  out[i >> 1] = (out[i >> 1] & ~(0xFFFFu << sh)) | (((uint32_t)x | ((uint32_t)y << 6)) << sh);
}
void mgo_aux_words(const mgo_env* e, uint32_t* out /* [4] */) {
  for (int i = 0; i < 3; ++i) out[i] = 0xFFFFFFFFu;
  out[3] = 0;
  if (e->kind == KIND_MEETUP) {
    for (int i = 0; i < e->n_goals; ++i) aux_put_goal(out, i, e->goal_x[i], e->goal_y[i]);
    for (int a = 0; a < e->A && a < 4; ++a) out[3] |= (uint32_t)e->past_dist[a] << (8 * a);
  }
  if ((e->kind == KIND_CLUTTERED && e->n_goals) || e->kind == KIND_FOURROOMS) {   /* the scattered goal, which the v8 shaper measures to */
    int gx, gy;
    goal_position(e, &gx, &gy);
    aux_put_goal(out, 0, gx, gy);
  }
}
/* what the last step/reset drew: order[A], respawn[A][4], spawn[A][4] */
void mgo_last_draws(const mgo_env* e, uint8_t* order, int8_t* respawn, int8_t* spawn) {
  if (order) memcpy(order, e->last_order, (size_t)e->A);
  if (respawn) memcpy(respawn, e->last_respawn, (size_t)e->A * 4);
  if (spawn) memcpy(spawn, e->last_spawn, (size_t)e->A * 4);
}
uint32_t mgo_errors(mgo_env* e) { uint32_t r = e->err; e->err = 0; return r; }
void mgo_get_grid(const mgo_env* e, uint8_t* cells /* [W*H*3], None -> (1,0,0) */) {
  for (int y = 0; y < e->H; ++y)
    for (int x = 0; x < e->W; ++x) {
      obj_t o = grid_get(&e->grid, x, y);
      uint8_t* p = cells + (y * e->W + x) * 3;
      if (o.type == T_NONE) { p[0] = T_EMPTY; p[1] = p[2] = 0; } else { p[0] = enc_type(o); p[1] = o.color; p[2] = o.state; }
    }
}
/* test hook: put an object on the live grid (used by DoorKey-style known-answer tests) */
void mgo_put_obj(mgo_env* e, int x, int y, int type, int color, int state) {
  obj_t o = {(uint8_t)type, (uint8_t)color, (uint8_t)state, 0};
  if (type == T_EMPTY) o = NONE;
  grid_set(&e->grid, x, y, o);
}
void mgo_set_step_count(mgo_env* e, int s) { e->step_count = s; }

void mgo_batch_state(mgo_env** envs, int64_t n, int64_t stride, uint32_t* agent_state /* [A][stride] */,
                     uint32_t* step_state /* [stride] */) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    for (int a = 0; a < envs[i]->A; ++a) agent_state[(int64_t)a * stride + i] = mgo_agent_state(envs[i], a);
    step_state[i] = mgo_step_state(envs[i]);
  }
}

/* ------------------------------------------------------------------------------------------
 * Batched drivers over the product's agent-major layouts (include/mgrid.h), so tests can
 * compare device arrays with oracle arrays element for element.  stride = env_stride.
 * ---------------------------------------------------------------------------------------- */
void mgo_batch_reset(mgo_env** envs, int64_t n, int64_t stride, const uint8_t* mask,
                     const int8_t* spawn /* [A][stride][4] or NULL */, uint64_t t,
                     uint8_t* image, uint8_t* dir) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    mgo_env* e = envs[i];
    if (mask && !mask[i]) continue;
    int8_t sp[MAXA * 4];
    if (spawn) for (int a = 0; a < e->A; ++a) memcpy(sp + 4 * a, spawn + ((int64_t)a * stride + i) * 4, 4);
    do_reset(e, spawn ? sp : NULL, t);
    if (image && dir) {
      uint8_t img[MAXA * MAXW * MAXW * 3], d[MAXA];
      int cb = mgo_obs_bytes(e);
      mgo_get_obs(e, img, d);
      for (int a = 0; a < e->A; ++a) {
        memcpy(image + ((int64_t)a * stride + i) * cb, img + a * cb, (size_t)cb);
        dir[(int64_t)a * stride + i] = d[a];
      }
    }
  }
}

void mgo_batch_step(mgo_env** envs, int64_t n, int64_t stride, const uint8_t* actions /* [A][stride] or NULL -> Philox */,
                    const uint8_t* order, const int8_t* respawn, const int8_t* reset_spawn, uint64_t t,
                    uint8_t* image, uint8_t* dir, double* reward, double* env_rew, uint8_t* done,
                    uint8_t* actions_out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    mgo_env* e = envs[i];
    int A = e->A, act[MAXA], cb = mgo_obs_bytes(e);
    uint8_t ord[MAXA]; int8_t rs[MAXA * 4], sp[MAXA * 4];
    if (actions) for (int a = 0; a < A; ++a) act[a] = actions[(int64_t)a * stride + i];
    else mgo_random_actions(e, t, act);
    if (actions_out) for (int a = 0; a < A; ++a) actions_out[(int64_t)a * stride + i] = (uint8_t)act[a];
    if (order) for (int k = 0; k < A; ++k) ord[k] = order[(int64_t)k * stride + i];
    if (respawn) for (int a = 0; a < A; ++a) memcpy(rs + 4 * a, respawn + ((int64_t)a * stride + i) * 4, 4);
    if (reset_spawn) for (int a = 0; a < A; ++a) memcpy(sp + 4 * a, reset_spawn + ((int64_t)a * stride + i) * 4, 4);
    uint8_t img[MAXA * MAXW * MAXW * 3], d[MAXA], dn;
    double rw[MAXA], er[MAXA];
    mgo_step(e, act, order ? ord : NULL, respawn ? rs : NULL, reset_spawn ? sp : NULL, t,
             image ? img : NULL, image ? d : NULL, rw, er, &dn);
    for (int a = 0; a < A; ++a) {
      int64_t o = (int64_t)a * stride + i;
      if (image) { memcpy(image + o * cb, img + a * cb, (size_t)cb); dir[o] = d[a]; }
      if (reward) reward[o] = rw[a];
      if (env_rew) env_rew[o] = er[a];
    }
    if (done) done[i] = dn;
  }
}

/* dynamic-grid kinds: batched drivers with the extra injection streams (same layouts as the
 * product: gen int8[stride][4]; objdrop int8[A][stride][4] agent-major like respawn) */
void mgo_batch_reset_ex(mgo_env** envs, int64_t n, int64_t stride, const uint8_t* mask, const int8_t* spawn,
                        const int8_t* gen, uint64_t t, uint8_t* image, uint8_t* dir) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    mgo_env* e = envs[i];
    if (mask && !mask[i]) continue;
    int8_t sp[MAXA * 4];
    if (spawn) for (int a = 0; a < e->A; ++a) memcpy(sp + 4 * a, spawn + ((int64_t)a * stride + i) * 4, 4);
    mgo_reset_gen(e, spawn ? sp : NULL, gen ? gen + i * 4 : NULL, t);
    if (image && dir) {
      uint8_t img[MAXA * MAXW * MAXW * 3], d[MAXA];
      int cb = mgo_obs_bytes(e);
      mgo_get_obs(e, img, d);
      for (int a = 0; a < e->A; ++a) {
        memcpy(image + ((int64_t)a * stride + i) * cb, img + a * cb, (size_t)cb);
        dir[(int64_t)a * stride + i] = d[a];
      }
    }
  }
}

void mgo_batch_step_ex(mgo_env** envs, int64_t n, int64_t stride, const uint8_t* actions, const uint8_t* order,
                       const int8_t* respawn, const int8_t* objdrop, const int8_t* reset_spawn, const int8_t* reset_gen,
                       uint64_t t, uint8_t* image, uint8_t* dir, double* reward, double* env_rew, uint8_t* done,
                       uint8_t* actions_out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    mgo_env* e = envs[i];
    int A = e->A, act[MAXA], cb = mgo_obs_bytes(e);
    uint8_t ord[MAXA]; int8_t rs[MAXA * 4], sp[MAXA * 4], od[MAXA * 4];
    if (actions) for (int a = 0; a < A; ++a) act[a] = actions[(int64_t)a * stride + i];
    else mgo_random_actions(e, t, act);
    if (actions_out) for (int a = 0; a < A; ++a) actions_out[(int64_t)a * stride + i] = (uint8_t)act[a];
    if (order) for (int k = 0; k < A; ++k) ord[k] = order[(int64_t)k * stride + i];
    if (respawn) for (int a = 0; a < A; ++a) memcpy(rs + 4 * a, respawn + ((int64_t)a * stride + i) * 4, 4);
    if (objdrop) for (int a = 0; a < A; ++a) memcpy(od + 4 * a, objdrop + ((int64_t)a * stride + i) * 4, 4);
    if (reset_spawn) for (int a = 0; a < A; ++a) memcpy(sp + 4 * a, reset_spawn + ((int64_t)a * stride + i) * 4, 4);
    uint8_t img[MAXA * MAXW * MAXW * 3], d[MAXA], dn;
    double rw[MAXA], er[MAXA];
    mgo_step_ex(e, act, order ? ord : NULL, respawn ? rs : NULL, objdrop ? od : NULL, reset_spawn ? sp : NULL,
                reset_gen ? reset_gen + i * 4 : NULL, t, image ? img : NULL, image ? d : NULL, rw, er, &dn);
    for (int a = 0; a < A; ++a) {
      int64_t o = (int64_t)a * stride + i;
      if (image) { memcpy(image + o * cb, img + a * cb, (size_t)cb); dir[o] = d[a]; }
      if (reward) reward[o] = rw[a];
      if (env_rew) env_rew[o] = er[a];
    }
    if (done) done[i] = dn;
  }
}

/* carry: [stride] words, or [stride][2] with five agents (byte a of the pair = what agent a carries) */
void mgo_batch_dyn_state(mgo_env** envs, int64_t n, int64_t stride, uint32_t* carry,
                         uint8_t* kinds /* [stride][H*W] */) {
  (void)stride;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    if (envs[i]->A > 4) {
      carry[2 * i] = mgo_carry_word(envs[i]);
      carry[2 * i + 1] = 0;
      for (int a = 4; a < envs[i]->A; ++a) carry[2 * i + 1] |= (uint32_t)kind_of(envs[i]->carrying[a]) << (8 * (a - 4));
    } else carry[i] = mgo_carry_word(envs[i]);
    mgo_grid_kinds(envs[i], kinds + i * envs[i]->W * envs[i]->H);
  }
}

/* scatter kinds: batched reset with the scattered objects' accepted positions injected
 * (scatter int8[stride][n_obj][2], or NULL -> Philox), aux state and recorder getters */
void mgo_batch_reset_scatter(mgo_env** envs, int64_t n, int64_t stride, const uint8_t* mask, const int8_t* spawn,
                             const int8_t* scatter, int n_obj, uint64_t t, uint8_t* image, uint8_t* dir) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    mgo_env* e = envs[i];
    if (mask && !mask[i]) continue;
    int8_t sp[MAXA * 4];
    if (spawn) for (int a = 0; a < e->A; ++a) memcpy(sp + 4 * a, spawn + ((int64_t)a * stride + i) * 4, 4);
    mgo_reset_scatter(e, spawn ? sp : NULL, scatter ? scatter + i * n_obj * 2 : NULL, t);
    if (image && dir) {
      uint8_t img[MAXA * MAXW * MAXW * 3], d[MAXA];
      int cb = mgo_obs_bytes(e);
      mgo_get_obs(e, img, d);
      for (int a = 0; a < e->A; ++a) {
        memcpy(image + ((int64_t)a * stride + i) * cb, img + a * cb, (size_t)cb);
        dir[(int64_t)a * stride + i] = d[a];
      }
    }
  }
}

void mgo_batch_aux(mgo_env** envs, int64_t n, uint32_t* aux /* [n][4] */, int8_t* scatter /* [n][n_obj][2] or NULL */, int n_obj) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    mgo_aux_words(envs[i], aux + i * 4);
    if (scatter) memcpy(scatter + i * n_obj * 2, envs[i]->last_scatter, (size_t)n_obj * 2);
  }
}
