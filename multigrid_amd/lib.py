"""ctypes binding of include/mgrid.h.  Fails loudly when libmgrid.so is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmgrid.so')

MG_OK = 0
SHAPING_NONE, SHAPING_V8 = 0, 1
PATH_AUTO, PATH_GENERIC, PATH_TABLE = 0, 1, 2
ERR_BAD_ACTION, ERR_BAD_RESPAWN, ERR_BAD_ORDER, ERR_MISSING_DRAW, ERR_BAD_MAP = 1, 2, 4, 8, 16
KIND_STATIC, KIND_DOORKEY, KIND_CLUTTERED, KIND_MEETUP, KIND_FOURROOMS = 0, 1, 2, 3, 4
KINDS = {'static': KIND_STATIC, 'doorkey': KIND_DOORKEY, 'cluttered': KIND_CLUTTERED, 'meetup': KIND_MEETUP,
         'fourrooms': KIND_FOURROOMS}
GEN_LAVA = 1
GEN_TWO_ROOMS = 2
GEN_LAVAWALL = 4
FIXED_OFF, FIXED_ALWAYS, FIXED_AT_RESET = 0, 1, 2


def slots(n_agents):
  """MG_SLOTS(n_agents): bytes per env of the action / order buffers (include/mgrid.h)"""
  return 4 if n_agents <= 4 else 8


def kind_id(type_, color=0, state=0):
  """MG_KIND_ID (include/mgrid.h): one-byte cell kind of the mutable per-env grids."""
  if type_ in (0, 1, 10):
    return 0
  if type_ == 4:
    return 128 + color * 4 + state
  return type_ * 8 + color


class MgConfig(C.Structure):
  _fields_ = [
      ('width', C.c_int32), ('height', C.c_int32), ('n_agents', C.c_int32), ('view_size', C.c_int32),
      ('max_steps', C.c_int32), ('see_through_walls', C.c_int32), ('competitive', C.c_int32),
      ('shaping_kind', C.c_int32), ('auto_reset', C.c_int32), ('env_kind', C.c_int32),
      ('emit_env_reward', C.c_int32), ('kernel_path', C.c_int32),
      ('n_envs', C.c_int64), ('env_id_base', C.c_int64), ('seed', C.c_uint64),
      ('gen_n_goals', C.c_int32), ('gen_n_clutter', C.c_int32), ('gen_flags', C.c_int32), ('reward_dtype', C.c_int32),
  ]


class MgBuffers(C.Structure):
  _fields_ = [
      ('n_envs', C.c_int64), ('env_stride', C.c_int64), ('n_agents', C.c_int32), ('obs_cell_bytes', C.c_int32),
      ('obs_image', C.c_void_p), ('reward', C.c_void_p), ('env_reward', C.c_void_p), ('dir_done', C.c_void_p),
      ('dir_done_bytes', C.c_int32), ('reward_elem_bytes', C.c_int32), ('env_state', C.c_void_p),
      ('carry', C.c_void_p), ('grid', C.c_void_p), ('grid_words', C.c_int32), ('mutable_grid', C.c_int32),
      ('aux', C.c_void_p), ('obs_full', C.c_void_p), ('obs_full_bytes', C.c_int32), ('reserved2', C.c_int32),
  ]


class MgEpisodeStats(C.Structure):
  _fields_ = [('ep_return', C.c_void_p), ('ep_goal_touches', C.c_void_p), ('ep_len', C.c_void_p), ('last_return', C.c_void_p),
              ('last_goal_touches', C.c_void_p), ('last_len', C.c_void_p), ('ep_count', C.c_void_p)]


class MgRolloutOut(C.Structure):
  _fields_ = [('obs_image', C.c_void_p), ('reward', C.c_void_p), ('dir_done', C.c_void_p), ('actions', C.c_void_p)]


# name -> (restype, argtypes): every symbol include/mgrid.h declares
SYMBOLS = {
    'mg_create': (C.c_int, [C.POINTER(MgConfig), C.c_int, C.POINTER(C.c_void_p)]),
    'mg_destroy': (C.c_int, [C.c_void_p]),
    'mg_set_static_map': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_set_maps': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_set_fixed_spawn': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    'mg_reset': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_reset_inject': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'mg_step': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_step_replay': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mg_reset_inject_dyn': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mg_step_replay_dyn': (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    'mg_reset_inject_scatter': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mg_export_grid': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_step_random': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_rollout_random': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(MgRolloutOut)]),
    'mg_rollout': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(MgRolloutOut)]),
    'mg_enable_episode_stats': (C.c_int, [C.c_void_p, C.c_double, C.POINTER(MgEpisodeStats)]),
    'mg_get_buffers': (C.c_int, [C.c_void_p, C.POINTER(MgBuffers)]),
    'mg_set_stream': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mg_get_stream': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'mg_sync': (C.c_int, [C.c_void_p]),
    'mg_get_step_index': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    'mg_set_step_index': (C.c_int, [C.c_void_p, C.c_uint64]),
    'mg_get_reset_index': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    'mg_set_reset_index': (C.c_int, [C.c_void_p, C.c_uint64]),
    'mg_check_errors': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    'mg_get_goal': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'mg_get_launch_info': (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 4),
    'mg_step_noalloc': (C.c_int, [C.POINTER(MgConfig), C.c_int, C.c_void_p, C.c_uint64] + [C.c_void_p] * 7),
    'mg_debug_device_sqrt': (C.c_int, [C.c_int, C.c_int32, C.c_void_p]),
    'mg_debug_device_rewards': (C.c_int, [C.c_int, C.c_int32, C.c_void_p]),
    'mg_debug_vis_masks': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'mg_debug_cell_code': (C.c_int, [C.c_int32] + [C.POINTER(C.c_uint32)] * 4),
    'mg_last_error': (C.c_char_p, []),
    'mg_version': (C.c_char_p, []),
}

_LIB = None


class MgError(RuntimeError):
  pass


def load():
  global _LIB
  if _LIB is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError(
          'multigrid_amd: %s is missing. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
          '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
      fn = getattr(lib, name)   # AttributeError if the library does not export it
      fn.restype = res
      fn.argtypes = args
    _LIB = lib
  return _LIB


def check(rc):
  if rc != MG_OK:
    raise MgError('mgrid error %d: %s' % (rc, load().mg_last_error().decode()))
