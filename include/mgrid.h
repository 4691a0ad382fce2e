/*
 * mgrid.h — C ABI of the MI355X-native batched MultiGrid step path.
 *
 * The reference (cxh42/multigrid) is pure Python and has no FFI: its boundary is the
 * duck-typed gym-0.23 env object (`reset()` / `step(actions)`), see
 *   envs/gym_multigrid/multigrid.py:466-501  (MultiGridEnv.reset)
 *   envs/gym_multigrid/multigrid.py:963-995  (MultiGridEnv.step)
 *   envs/gym_multigrid/multigrid.py:1037-1083 (gen_obs / gen_agent_obs)
 *   v8_robust_ppo.py:171-220                 (v8 reward shaping, fused into step here)
 * Every entry point below names the reference call it replaces.  The library keeps N
 * independent envs resident in HBM as struct-of-arrays and advances all of them with one
 * kernel launch per call.
 *
 * Conventions
 *  - every function returns MG_OK (0) or a negative MG_E* code; never throws, never aborts.
 *    mg_last_error() returns the thread-local text of the last failure.
 *  - a handle is bound to one HIP device and one stream.  Calls on one handle are not
 *    re-entrant; different handles may be driven from different host threads.
 *  - reset/step calls are asynchronous on the handle's stream; mg_sync() blocks.
 *  - env_stride = mg_buffers.env_stride = n_envs rounded up to a multiple of 64; every array is
 *    allocated (by the library) or expected (from the caller) with env_stride env slots.
 *  - the big outputs are AGENT-MAJOR: obs_image and reward put element (agent a, env n) at
 *    a*env_stride + n, mirroring the reference's obs dict whose 'image' entry is a list indexed
 *    by agent first (multigrid.py:1060-1063) and giving each per-agent policy network a
 *    contiguous [N, ...] tensor.
 *  - small per-env records are ENV-MAJOR, one naturally aligned word per env so that a wavefront
 *    moves them with one coalesced instruction: actions (4 bytes), order (4 bytes), dir_done
 *    (4 or 8 bytes), env_state (A+1 dwords), respawn/spawn (A dwords).
 *  - the library owns all device buffers; mg_get_buffers() hands out borrowed device
 *    pointers that stay valid until mg_destroy().
 */
#ifndef MGRID_H
#define MGRID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes per env of the action and order buffers: one dword up to four agents, two dwords for five (the reference's
 * Empty-16x16, FourRooms-v0 and DoorKey-16x16 ids) */
#define MG_MAX_AGENTS 5
#define MG_SLOTS(n_agents) ((n_agents) <= 4 ? 4 : 8)

#define MG_OK            0
#define MG_EINVAL       -1  /* bad argument / config */
#define MG_EHIP         -2  /* HIP runtime error (text in mg_last_error) */
#define MG_ENOMEM       -3
#define MG_EUNSUPPORTED -4  /* config outside what the kernels were instantiated for */
#define MG_ESTATE       -5  /* call order (e.g. step before map/reset) */

/* shaping_kind */
#define MG_SHAPING_NONE 0   /* reward = env reward (multigrid.py:881-883) */
#define MG_SHAPING_V8   1   /* reward = v8_robust_ppo.py:171-220 shaped reward */
#define MG_SHAPING_V7   2   /* reward = v7_stable_goal_focused_ppo.py:60-108 shaped reward (StableRewardShaper) */

/* reward_dtype */
#define MG_REWARD_F64 0
#define MG_REWARD_F32 1

/* kernel_path: how the static part of an agent's view is produced.  TABLE = precomputed once
 * per (free cell, dir) on the host and kept in LDS; GENERIC = slice/rotate/process_vis evaluated
 * per step from the LDS map.  Both give identical bytes.  AUTO: GENERIC below 65,536 envs (no table fill), TABLE with
 * persistent workgroups up to four chunks per resident wave (1,048,576 envs on 256 CUs), GENERIC launched one chunk per
 * wave beyond (in-order short-lived workgroups write faster once the working set exceeds the Infinity Cache).
 * Independently of the path, a launch beyond the resident batch (and a mutable-grid launch that writes 192 MiB of obs or
 * more) writes its obs images with non-temporal stores, a resident table launch of that size those of all agents but
 * the last; the bytes are the same. */
#define MG_PATH_AUTO    0
#define MG_PATH_GENERIC 1
#define MG_PATH_TABLE   2

/* env_kind: what _gen_grid does at every reset.
 *   STATIC  = one fixed layout shared by all envs (mg_set_static_map), agents spawn anywhere
 *             (ClutteredMultiGridFixed15x15, cluttered.py:47-63).  A layout of only empty/wall/goal/lava
 *             cells runs the shared-map kernels; a layout that also holds keys, balls, boxes, doors or
 *             floor tiles gives every env its own mutable copy (the "mutable-grid" kernel below).
 *   DOORKEY = DoorKeyEnv._gen_grid (multigrid_envs/doorkey.py:34-86) drawn on device per env and per
 *             reset: border walls, goal at (W-2,H-2), a wall column at a random split with a locked
 *             yellow door, agents and a yellow key left of it.  No map upload; W == H >= 5.
 *   CLUTTERED = ClutteredMultiGrid._gen_grid (multigrid_envs/cluttered.py:25-40) drawn on device per env and
 *             per reset: border walls, a goal (scattered when gen_n_goals = 1 = randomize_goal, at (W-2,H-2)
 *             when 0), gen_n_clutter walls (lava with MG_GEN_LAVA) scattered by place_obj, agents anywhere.
 *   MEETUP  = MeetupEnv (multigrid_envs/meetup.py:26-95): gen_n_goals (1..6) locked red doors as meeting
 *             points and gen_n_clutter walls scattered, fully observed (obs = the whole encoded grid,
 *             mg_buffers.obs_full, multigrid.py:1043-1045), reward = decrease of the Manhattan distance to the
 *             jointly closest goal, +1 and done when every agent stands next to it (meetup.py:78-95). */
#define MG_KIND_STATIC    0
#define MG_KIND_DOORKEY   1
#define MG_KIND_CLUTTERED 2
#define MG_KIND_MEETUP    3
#define MG_KIND_FOURROOMS 4   /* FourRoomsEnv._gen_grid (two_rooms with MG_GEN_TWO_ROOMS), random agents and goal (multigrid_envs/
                                 fourrooms.py:53-104): walls at x = W/2 and y = H/2 with one random gap per segment, drawn per
                                 env and per reset; the goal is placed after the agents.  Injected as three scatter pairs:
                                 (gap_top_y, gap_left_x), (gap_right_x, gap_bottom_y), (goal_x, goal_y). */
#define MG_GEN_LAVA 1      /* gen_flags: Cluttered walls_are_lava */
#define MG_GEN_LAVAWALL 4  /* gen_flags: Cluttered obstacles are LavaWall objects (lava_walls.py:25-38, 79-80): deadly like lava,
                              transparent, but encoded as grey walls.  In a cell triple / kind id: lava with colour grey. */
#define MG_GEN_TWO_ROOMS 2 /* gen_flags: FourRooms two_rooms=True (fourrooms.py:75-83): only the vertical wall is built; its lower
                              gap is not drawn on grids with height < 7 */
#define MG_MAX_SCATTER 64  /* gen_n_goals + gen_n_clutter */

/* One-byte cell kind of the mutable per-env grids, as the C ABI speaks it (mg_export_grid, mg_buffers.carry, injected
 * layouts): 0 = nothing, door = 128 + color*4 + state, any other object = type*8 + color.  Agents are not
 * stored in the grid (their cells read 0): the reference keeps an Agent object in the cell, which
 * move_agent / agent_is_done always replace by None when the agent leaves (multigrid.py:817-847).
 * (Inside the library the grids hold a bit-field form of these ids — opaque | type<<1 | colour<<5, csrc/mg_dyn.h — so that
 * a view decodes a cell with one multiply; mg_buffers.grid is that raw storage, mg_export_grid translates.) */
#define MG_KIND_ID(type, color, state) ((type) == MG_T_DOOR ? 128 + (color) * 4 + (state) : \
                                        ((type) <= MG_T_EMPTY || (type) == MG_T_AGENT) ? 0 : (type) * 8 + (color))

/* cell encoding = gym-minigrid OBJECT_TO_IDX / COLOR_TO_IDX / door state */
#define MG_T_UNSEEN 0
#define MG_T_EMPTY  1
#define MG_T_WALL   2
#define MG_T_FLOOR  3
#define MG_T_DOOR   4
#define MG_T_KEY    5
#define MG_T_BALL   6
#define MG_T_BOX    7
#define MG_T_GOAL   8
#define MG_T_LAVA   9
#define MG_T_AGENT 10

/* bits of the device error word (mg_check_errors) */
#define MG_ERR_BAD_ACTION   1u  /* action outside [0,6]   (multigrid.py:902 assert) */
#define MG_ERR_BAD_RESPAWN  2u  /* injected respawn/spawn/object cell occupied or out of range, or bad injected layout draw */
#define MG_ERR_BAD_ORDER    4u  /* injected order is not a permutation */
#define MG_ERR_MISSING_DRAW 8u  /* replay step needed a respawn/spawn that was not injected */
#define MG_ERR_BAD_MAP     16u  /* mg_step_noalloc: the layout holds a cell type shared static maps cannot hold, or fewer free
                                   cells than agents (mg_set_static_map refuses both on the host; a device layout cannot be) */

/* Mirrors the ctor kwargs of MultiGridEnv (multigrid.py:325-339) + ClutteredMultiGrid
 * (cluttered.py:16-23), plus batching/sharding fields that have no reference counterpart. */
typedef struct mg_config {
  int32_t width;             /* grid width  W (multigrid.py:446) */
  int32_t height;            /* grid height H */
  int32_t n_agents;          /* A (multigrid.py:375) */
  int32_t view_size;         /* agent_view_size V, odd (multigrid.py:387) */
  int32_t max_steps;         /* multigrid.py:448 */
  int32_t see_through_walls; /* multigrid.py:449, 1020-1024 */
  int32_t competitive;       /* multigrid.py:378-381 (forced on when n_agents == 1) */
  int32_t shaping_kind;      /* MG_SHAPING_* */
  int32_t auto_reset;        /* 1: an env whose step returns done is reset inside the same
                                kernel and the obs written is the post-reset obs (gym vector
                                autoreset; v7_stable_goal_focused_ppo.py:334-340 is the only
                                reference precedent).  0: reference behaviour, caller resets. */
  int32_t env_kind;          /* MG_KIND_* */
  int32_t emit_env_reward;   /* 1: also write the unshaped env reward (mg_buffers.env_reward) */
  int32_t kernel_path;       /* MG_PATH_AUTO (default), or force MG_PATH_GENERIC / MG_PATH_TABLE */
  int64_t n_envs;            /* envs held by THIS handle (this GPU's shard) */
  int64_t env_id_base;       /* global id of local env 0; Philox streams are keyed by the
                                global id so results do not depend on the sharding */
  uint64_t seed;             /* Philox key for free-running mode */
  int32_t gen_n_goals;       /* CLUTTERED: 1 = randomize_goal (cluttered.py:28-31); MEETUP: n_goals (meetup.py:32) */
  int32_t gen_n_clutter;     /* n_clutter (cluttered.py:16, meetup.py:33) */
  int32_t gen_flags;         /* MG_GEN_* */
  int32_t reward_dtype;      /* MG_REWARD_F64 (0, default: the reference's Python floats) or MG_REWARD_F32: the reward buffers
                                hold float, 12 bytes less to write per env-step with three agents.  A bandwidth option OUTSIDE the
                                reference's precision: v8 runs compute_gae on the float64 rewards and casts only the resulting
                                advantages and returns to float32 (v8_robust_ppo.py:304-311), so rounding the rewards first is
                                not its arithmetic.  Not with emit_env_reward. */
} mg_config;

typedef struct mg_buffers {
  int64_t  n_envs;
  int64_t  env_stride;   /* elements between consecutive agents' rows (multiple of 64) */
  int32_t  n_agents;
  int32_t  obs_cell_bytes; /* V*V*3 : bytes of one agent's image, [vx][vy][(type,color,state)]
                              exactly as Grid.encode lays it out */
  uint8_t* obs_image;    /* [A][env_stride][V*V*3]  — obs['image'][a]   */
  double*  reward;       /* [A][env_stride]  rewards[a] (shaped when shaping_kind != NONE); float[A][env_stride] behind the same
                            pointer when reward_dtype = MG_REWARD_F32 (reward_elem_bytes = 4) */
  double*  env_reward;   /* [A][env_stride]  unshaped env reward, NULL unless emit_env_reward */
  uint8_t* dir_done;     /* [env_stride][dir_done_bytes]: byte a = obs['direction'][a] (a < A), byte A =
                            collective done of the step just taken; dir_done_bytes = 4 (A <= 3) or 8 */
  int32_t  dir_done_bytes;
  int32_t  reward_elem_bytes;   /* 8, or 4 with MG_REWARD_F32 */
  uint32_t* env_state;   /* [env_stride][A+1]: words 0..A-1 = one word per agent
                              x:6 | y:6<<6 | dir:2<<12 | prev_x:6<<14 | prev_y:6<<20 | valid<<26 | stationary:3<<27
                            (post-step, post-respawn position = what v8 reads as env.agent_pos[a], v8:242;
                            prev_x, prev_y, valid, stationary = v8 shaper state, v8:127-129, the count saturating at 7
                            because only "> 3" is ever tested, v8:202); word A = step_count | agent done bits<<16
                            | (DOORKEY only) split_idx<<24 (6 bits), the width of the region agents (re)spawn in
                            (doorkey.py:77-78), and the low five bits of door_idx, the door's row (doorkey.py:58-61), in
                            bits 21..23 (its bits 0..2) and 30..31 (its bits 3..4) */
  /* mutable-grid handles only (mutable_grid != 0), NULL / 0 otherwise: */
  uint32_t* carry;       /* [env_stride] (five agents: [env_stride][2]): byte a = kind id of the object agent a carries,
                            0 = nothing (multigrid.py:490, 926-951) */
  uint32_t* grid;        /* dwords [env_stride/64][grid_words/4][64][4]: the library's cell codes (see MG_KIND_ID), cell-major
                            within a 64-env chunk.  With l = n%64, D = 16*ci + l/4, k = D/64, L = D%64, the code of cell
                            ci = y*W + x of env n is byte l%4 of dword ((n/64)*(grid_words/4) + k/4)*256 + L*4 + k%4
                            (cells >= W*H are padding): a wavefront moves its chunk between HBM and LDS as fully
                            coalesced 16-byte vectors.  Use mg_export_grid for a plain [n][H*W] copy. */
  int32_t  grid_words;   /* dwords per env in `grid`: ceil(W*H/4) rounded up to a multiple of 4 */
  int32_t  mutable_grid;
  uint32_t* aux;         /* MEETUP, CLUTTERED, FOURROOMS, [env_stride][4]: words 0..2 = MeetupEnv.goal_pos, two per word
                            (goal 2k in bits 0..15 of word k, goal 2k+1 in bits 16..31, each x | y<<6, 0xFFFF if unused:
                            up to six goals), word 3 = past_goal_dist, one byte per agent (meetup.py:61,70,94);
                            CLUTTERED with a scattered goal / FOURROOMS: bits 0..15 of word 0 = the env's goal cell (what
                            the v8 shaper's get_goal_position finds, v8_robust_ppo.py:159-169) */
  uint8_t* obs_full;     /* fully observed kinds (MEETUP) only, [env_stride][W*H*3]: Grid.encode() of the whole grid,
                            [x][y][(type,color,state)] with the agents in their cells as (10, id, dir); the same image
                            for every agent (multigrid.py:1043-1045), obs['position'][a] is in env_state.  obs_image is
                            NULL for these kinds. */
  int32_t  obs_full_bytes;
  int32_t  reserved2;
} mg_buffers;

typedef struct mg_handle mg_handle;

/* gym.make(env_id) + ctor (v8_robust_ppo.py:124; multigrid.py:325-464).  Allocates all
 * device state for cfg->n_envs envs on `device`.  No map yet: call mg_set_static_map. */
int mg_create(const mg_config* cfg, int device, mg_handle** out);

/* env.close() (v8_robust_ppo.py:383). */
int mg_destroy(mg_handle* h);

/* Result of _gen_grid's static part (cluttered.py:25-36): one layout shared by every env.
 * cells = host pointer, uint8[W*H*3] indexed [(y*W+x)*3 + c] = (type,color,state).
 * With only empty/wall/goal/lava cells the map stays shared and immutable (nothing pickable, nothing
 * toggleable, nothing an agent can walk onto and destroy — multigrid.py:842-845,926-961).  Any key,
 * ball, box (empty), door or floor cell switches the handle to the mutable-grid kernel: every reset
 * copies the layout into the env's own grid (mg_buffers.grid) and step applies pickup / drop / toggle /
 * overlap to it (multigrid.py:926-961, 112-122, 834-847).  MG_EINVAL on a generated kind (no map).  A new map
 * invalidates the envs' state: mg_reset must be called again before the next step (MG_ESTATE otherwise). */
int mg_set_static_map(mg_handle* h, const uint8_t* cells);

/* Per-env layouts (SURVEY §8(b)'s mg_set_maps): env n resets to its OWN layout instead of one shared by all — the batched form of
 * constructing N reference envs with N different fixed maps.  cells = DEVICE pointer uint8[n_envs][H*W*3], each env's layout as in
 * mg_set_static_map.  The handle switches to the mutable-grid kernel (every env owns its grid; objects allowed), each env's v8
 * shaper goal is its own first goal cell (mg_buffers.aux word 0, bits 0..15), agents spawn anywhere.  Unknown cell types are
 * imported as walls and reported as MG_ERR_BAD_MAP (mg_check_errors), as are layouts with fewer free cells than agents.
 * MG_KIND_STATIC handles only; mg_reset is required afterwards; a later mg_set_static_map goes back to one shared layout.
 * mg_get_buffers must be called again afterwards (carry / grid / aux appear).  cells is read by a kernel on the handle's stream:
 * keep it alive and unchanged until that stream has passed the call (mg_sync). */
int mg_set_maps(mg_handle* h, const uint8_t* cells);

/* Episode statistics kept by the library: what run_episode keeps beside the trajectories (v8_robust_ppo.py:234-235, 246-248,
 * 266, 273 — episode_rewards[a] += shaped_rewards[a]; goal_touches += 1 for every shaped reward >= 4.5) and what train() logs
 * per finished episode (v8_robust_ppo.py:425), per env, on the device.  Off by default; once enabled every mg_reset* / mg_step* /
 * mg_rollout_random call is followed, on the handle's stream, by one small kernel that folds that call's rewards and done flags
 * in (about 30 A + 40 bytes of HBM traffic per env-step, one thread per env; a rollout's K steps are folded in one pass over
 * its trajectory).  Sums are added per env in step order in float64: bit-identical to the reference's Python floats.
 * touch_threshold: a reward >= it counts as a goal touch (4.5 for the v8 shaper; any positive value below 0.1 counts unshaped
 * goal rewards, multigrid.py:879-886).  All pointers are DEVICE memory owned by the handle, agent-major like mg_buffers.reward. */
typedef struct mg_episode_stats {
  double*   ep_return;         /* [A][env_stride] return of the episode in progress */
  uint32_t* ep_goal_touches;   /* [env_stride]    its goal touches so far */
  uint32_t* ep_len;            /* [env_stride]    its steps so far */
  double*   last_return;       /* [A][env_stride] the most recently finished episode of this env: return, */
  uint32_t* last_goal_touches; /* [env_stride]    goal touches, */
  uint32_t* last_len;          /* [env_stride]    length; valid once ep_count > 0 */
  uint32_t* ep_count;          /* [env_stride]    episodes finished (done reported) since mg_enable_episode_stats */
} mg_episode_stats;
int mg_enable_episode_stats(mg_handle* h, double touch_threshold, mg_episode_stats* out);

/* Fixed start cells on a static map (shared-map kernels only).  xy = host int8[A][2].
 *   MG_FIXED_ALWAYS   EmptyEnv(agent_start='fixed').place_one_agent (multigrid_envs/empty.py:33-36, 72-104): every
 *                     placement — reset and the respawn after a goal / lava touch — puts the agent on its start cell
 *                     facing east; an agent found on that cell is sent back to its own start cell first (and so on).
 *   MG_FIXED_AT_RESET MazeEnv._gen_grid (multigrid_envs/maze.py:86): reset places the agent on its start cell with a
 *                     drawn direction; respawns stay random (multigrid.py:832).
 *   MG_FIXED_OFF      back to random placement.
 * A later mg_set_static_map switches it off again; mg_reset is required afterwards. */
#define MG_FIXED_OFF      0
#define MG_FIXED_ALWAYS   1
#define MG_FIXED_AT_RESET 2
int mg_set_fixed_spawn(mg_handle* h, const int8_t* xy, int mode);

/* env.reset() (multigrid.py:466-501) for every env whose mask byte is non-zero (mask = device
 * pointer uint8[n_envs], or NULL for all).  Spawns are drawn on device (Philox) with the
 * reference's distribution: uniform rejection sampling over the grid for an empty cell not
 * holding an agent, then uniform direction (multigrid.py:598-619, 665-666).  Writes obs.
 * Every explicit reset call draws afresh (Philox counter word = bit 63 | number of mg_reset* calls on the handle so far):
 * reset(); reset() gives new spawns / layouts, as the reference's clock-seeded place_agent does (cluttered.py:55-63), and
 * an explicit reset never repeats the draws of the auto-reset inside a step. */
int mg_reset(mg_handle* h, const uint8_t* env_mask);

/* Same, with the spawn the reference's time-seeded RNG would have produced injected:
 * spawn = device pointer int8[env_stride][A][4] = (x, y, dir, 0) (cluttered.py:55-63). */
int mg_reset_inject(mg_handle* h, const uint8_t* env_mask, const int8_t* spawn);

/* env.step(actions) (multigrid.py:963-995) for all envs, free-running: the agent order
 * (multigrid.py:973-974) and respawns (multigrid.py:815-832) are drawn on device (Philox).
 * actions = device pointer uint8[env_stride][MG_SLOTS(n_agents)]: byte a of env n = action of agent a in [0,6]
 * (the list `env.step(actions)` takes, multigrid.py:963-978); bytes >= A are ignored. */
int mg_step(mg_handle* h, const void* actions);

/* env.step(actions) with the reference's two RNG streams injected (bit-exact replay):
 *   order   = device uint8[env_stride][MG_SLOTS(n_agents)]: byte k of env n = agent acting k-th (the shuffled
 *             np.arange of multigrid.py:973-974)
 *   respawn = device int8[env_stride][A][4] = (x,y,dir,0) accepted by place_one_agent for
 *             agent a if it finishes this step (multigrid.py:832), x = -1 if none recorded
 *   reset_spawn = same layout, used when auto_reset fires; may be NULL when auto_reset == 0 */
int mg_step_replay(mg_handle* h, const void* actions, const uint8_t* order,
                   const int8_t* respawn, const int8_t* reset_spawn);

/* Mutable-grid replay (DoorKey / maps with objects): mg_reset_inject / mg_step_replay plus the two
 * extra draw streams those envs have.  Any of the extra pointers may be NULL (drawn on device then).
 *   gen / reset_gen = device int8[env_stride][4] = (split_idx, door_idx, key_x, key_y): the layout draws of
 *             DoorKeyEnv._gen_grid (doorkey.py:51,59,64-65) for a reset / for the auto-reset of this step
 *   objdrop = device int8[env_stride][A][4] = (x, y, 0, 0): where place_obj put the object agent a carried
 *             when it finished this step (multigrid.py:827-829), x = -1 if none recorded */
int mg_reset_inject_dyn(mg_handle* h, const uint8_t* env_mask, const int8_t* spawn, const int8_t* gen);
int mg_step_replay_dyn(mg_handle* h, const void* actions, const uint8_t* order, const int8_t* respawn,
                       const int8_t* objdrop, const int8_t* reset_spawn, const int8_t* reset_gen);

/* Reset of the scatter kinds (CLUTTERED, MEETUP; FOURROOMS with its three pairs) with the layout the reference drew
 * injected: scatter = device int8[env_stride][gen_n_goals + gen_n_clutter][2] = the (x, y) place_obj accepted for every scattered object in
 * draw order (goal / goal doors first, then the clutter; cluttered.py:28-36, meetup.py:67-72); spawn as in
 * mg_reset_inject.  Either pointer may be NULL (drawn on device).  Auto-resets inside step always draw on device. */
int mg_reset_inject_scatter(mg_handle* h, const uint8_t* env_mask, const int8_t* spawn, const int8_t* scatter);

/* env.grid as plain per-env arrays (multigrid.py:281-319 Grid.grid minus the Agent objects):
 * out = device uint8[n_envs][H*W], out[n][y*W+x] = kind id.  Mutable-grid handles only. */
int mg_export_grid(mg_handle* h, uint8_t* out);

/* Synthetic-rollout step: like mg_step but the actions are drawn on device too
 * (uniform in [0,6], the "random actions" of BASELINE.json configs) and also stored to
 * `actions_out` (device uint8[env_stride][MG_SLOTS(n_agents)], same layout as `actions`; may be NULL). */
int mg_step_random(mg_handle* h, uint8_t* actions_out);

/* mg_step for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (cxh42/multigrid is pure Python): it exists
 * for callers that must not reach hipMalloc — code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation.  One free-running env.step(actions) (multigrid.py:963-995 + v8 shaping + auto-reset, as mg_step) of cfg->n_envs
 * envs on a shared static layout, with NO handle: every buffer is the caller's, and what a handle keeps in hipMalloc'ed
 * tables (padded map, goal-distance table, error word) lives in a 64 KB file-scope __device__ array of the library, one copy
 * per GPU, held until the process ends.  The call itself allocates and frees nothing and does not synchronise: it enqueues
 * one 4-byte memset, a one-workgroup kernel that rebuilds the tables from `cells`, the step kernel and (err_out != NULL) one
 * 4-byte device-to-device copy on `hip_stream`.  The array's device address is looked up once per device
 * (hipGetSymbolAddress), by the first mg_create on that device or else by the first call of this function there.
 *   cfg        as for mg_create; env_kind must be MG_KIND_STATIC, emit_env_reward 0; kernel_path is ignored (the generic
 *              kernel, one chunk per wave); seed / env_id_base key the Philox draws exactly as in mg_step
 *   step_index the Philox step index of this step (what mg_get_step_index reports for a handle)
 *   cells      DEVICE uint8[H*W*3], the layout mg_set_static_map takes from the host: empty / wall / goal / lava cells only
 *              (anything else is treated as a wall and reported as MG_ERR_BAD_MAP)
 *   actions    device uint8[env_stride][MG_SLOTS(A)]                       (in)
 *   env_state  device uint32[env_stride][A+1], layout of mg_buffers        (in/out)
 *   obs_image  device uint8[A][env_stride][V*V*3]; reward device double[A][env_stride] — float[A][env_stride] behind the same
 *              pointer when cfg->reward_dtype = MG_REWARD_F32; dir_done device uint8[env_stride][4 or 8]
 *                                                                          (out), env_stride = n_envs rounded up to 64
 *   err_out    device uint32 (may be NULL): the MG_ERR_* bits of this call
 * Limits: width, height <= 64 and view_size in {3, 5, 7} (what the array was laid out for; MG_EINVAL / MG_EUNSUPPORTED
 * beyond), any n_envs.  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two
 * threads) race on it — serialise them. */
int mg_step_noalloc(const mg_config* cfg, int device, void* hip_stream, uint64_t step_index, const uint8_t* cells,
                    const void* actions, uint32_t* env_state, uint8_t* obs_image, double* reward, uint8_t* dir_done,
                    uint32_t* err_out);

/* Fused synthetic rollout: n_steps consecutive mg_step_random steps in ONE kernel launch (the "synthetic random-action
 * rollouts" BASELINE.json's metric is quoted on; the reference's counterpart is a loop of env.step(random actions),
 * v8_robust_ppo.py:237-271 with the policy replaced by a uniform draw).  Each wavefront keeps its 64 envs' state in
 * registers over the n_steps steps; step k's outputs go to slot k of the caller's trajectory buffers, laid out like the
 * handle's own (device pointers, env_stride and dir_done_bytes as in mg_buffers):
 *   obs_image uint8 [n_steps][A][env_stride][V*V*3]     reward double [n_steps][A][env_stride]
 *   dir_done  uint8 [n_steps][env_stride][dir_done_bytes]   actions uint8 [n_steps][env_stride][MG_SLOTS(A)] (may be NULL)
 * Afterwards the handle is exactly where n_steps calls of mg_step_random would have left it: env_state, step index, and
 * its own obs_image / reward / dir_done buffers hold the last step.  Handles without env_reward output only (MG_EUNSUPPORTED
 * otherwise); auto-reset and shaping as configured.  Mutable-grid handles (DoorKey, Meetup, drawn layouts, layouts with objects,
 * mg_set_maps) take the same call with the same result and, since round 3, their own fused kernel: a wavefront copies its 64
 * envs' grids into LDS once, takes them through the n_steps steps there (layout generation at auto-resets, pickups, doors), and
 * writes grids / env_state / carry back once; fully observed kinds: obs_image is uint8 [n_steps][env_stride][W*H*3]
 * (mg_buffers.obs_full). */
typedef struct mg_rollout_out {
  uint8_t* obs_image;
  double*  reward;
  uint8_t* dir_done;
  uint8_t* actions;
} mg_rollout_out;
int mg_rollout_random(mg_handle* h, int32_t n_steps, const mg_rollout_out* out);

/* The same fused rollout with the CALLER's actions: step k takes slot k of `actions`, device uint8
 * [n_steps][env_stride][MG_SLOTS(A)] laid out like mg_step's (the loop env.step(actions_k) for k in range(K) over a recorded or
 * open-loop action sequence, v8_robust_ppo.py:237-239); order and respawn draws stay on the device (free-running).  The handle ends
 * exactly where n_steps calls of mg_step would have left it.  out->actions is not written.  actions == NULL: mg_rollout_random. */
int mg_rollout(mg_handle* h, int32_t n_steps, const void* actions, const mg_rollout_out* out);

int mg_get_buffers(mg_handle* h, mg_buffers* out);

/* Stream control: by default the handle owns a stream; mg_set_stream adopts the caller's
 * (e.g. torch's current stream) — pass a hipStream_t cast to void*. */
int mg_set_stream(mg_handle* h, void* hip_stream);
int mg_get_stream(mg_handle* h, void** hip_stream);
int mg_sync(mg_handle* h);

/* Number of steps taken so far (the Philox step index of the next step). */
int mg_get_step_index(mg_handle* h, uint64_t* out);
int mg_set_step_index(mg_handle* h, uint64_t t);

/* Number of explicit reset CALLS made so far (mg_reset, mg_reset_inject*; masked or injected, each call counts once for
 * every env of the handle): an explicit reset draws under the Philox counter word 2^63 | this number in place of the step
 * index, so that reset(); reset() redraws (the reference reseeds from the clock, cluttered.py:55-63) and never repeats the
 * draws of an auto-reset inside a step.  Together with the step index it is all of a handle's generator state: a resumed or
 * re-created handle (checkpoint, shard change) that restores both reproduces the draws of the one it replaces. */
int mg_get_reset_index(mg_handle* h, uint64_t* out);
int mg_set_reset_index(mg_handle* h, uint64_t n);

/* Synchronises, returns and clears the device error word (MG_ERR_* bits). */
int mg_check_errors(mg_handle* h, uint32_t* out);

/* Goal cell the v8 shaper measures distance to (v8_robust_ppo.py:159-169: first 'goal' cell
 * scanning x-major, (13,13) if none). */
int mg_get_goal(mg_handle* h, int32_t* gx, int32_t* gy);

/* Kernel path actually in use (MG_PATH_GENERIC / MG_PATH_TABLE) and the launch geometry of a plain
 * mg_step on this handle (a small batch: the kernel that deals a chunk's agents out over waves, as
 * many waves per workgroup as agents); any out pointer may be NULL.  No reference counterpart
 * (diagnostics for DESIGN.md / bench.py). */
int mg_get_launch_info(mg_handle* h, int32_t* path, int32_t* block_threads, int32_t* grid_blocks, int32_t* lds_bytes);

/* Test hook (no reference counterpart): MiniGridEnv._reward() = 1 - 0.9*(s/max_steps) as the
 * DEVICE computes it, for s = 0..max_steps, copied to host `out` (max_steps+1 doubles). */
int mg_debug_device_rewards(int device, int32_t max_steps, double* out);

/* Test hook (no reference counterpart): sqrt(k), k = 0..n, as mg_step_noalloc's table-building kernel evaluates it on the
 * device (np.linalg.norm of an integer vector, v8_robust_ppo.py:192), copied to host `out` (n+1 doubles). */
int mg_debug_device_sqrt(int device, int32_t n, double* out);

/* Test hook, host only (no GPU call, no reference counterpart): evaluates the kernels' bitmask
 * form of Grid.process_vis on n opacity patterns so that it can be compared
 * exhaustively with the oracle's literal loops.  V in {3,5,7}; bit (i*V+j) of a pattern = view
 * cell (vx=i, vy=j) is opaque; `out` uses the same bit layout for "visible". */
int mg_debug_vis_masks(int V, const uint64_t* patterns, int64_t n, uint64_t* out);

/* Test hook, host only (no GPU call, no reference counterpart): how the mutable grids store public kind id `kind`
 * (MG_KIND_ID): *code = the one-byte cell code, *kind_back = the id mg_export_grid turns it back into, *table_word =
 * encode() triple (type | colour<<8 | state<<16) | behaviour flags<<24 as the kernels' table holds it, *decoded =
 * type | colour<<8 as the views' arithmetic decode extracts it from the code. */
int mg_debug_cell_code(int32_t kind, uint32_t* code, uint32_t* kind_back, uint32_t* table_word, uint32_t* decoded);

const char* mg_last_error(void);
const char* mg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MGRID_H */
