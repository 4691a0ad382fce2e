"""mg_enable_episode_stats through the C ABI: the per-episode bookkeeping of run_episode (/root/reference/v8_robust_ppo.py:234-235,
246-248, 266, 273: episode_rewards[a] += shaped_rewards[a]; goal_touches += 1 per shaped reward >= 4.5; what train() logs at :425),
restated here as plain per-step NumPy accumulation over the rewards / done flags the device returned — bit-exact float64 sums."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _need_gpu():
  import torch
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


class RefStats:
  """run_episode's accumulators for n envs at once, one step at a time, in the reference's order of additions."""

  def __init__(self, A, n, thr):
    self.thr = thr
    self.ret = np.zeros((A, n)); self.touch = np.zeros(n, np.int64); self.len = np.zeros(n, np.int64)
    self.last_ret = np.zeros((A, n)); self.last_touch = np.zeros(n, np.int64); self.last_len = np.zeros(n, np.int64)
    self.count = np.zeros(n, np.int64)

  def reset(self, mask=None):
    m = np.ones(self.len.shape, bool) if mask is None else np.asarray(mask, bool)
    self.ret[:, m] = 0.0; self.touch[m] = 0; self.len[m] = 0

  def step(self, reward, done):
    reward = np.asarray(reward, np.float64)
    for a in range(reward.shape[0]):           # episode_rewards[i] += shaped_rewards[i]
      self.ret[a] += reward[a]
      self.touch += reward[a] >= self.thr
    self.len += 1
    d = np.asarray(done, bool)
    self.last_ret[:, d] = self.ret[:, d]; self.last_touch[d] = self.touch[d]; self.last_len[d] = self.len[d]
    self.count += d
    self.ret[:, d] = 0.0; self.touch[d] = 0; self.len[d] = 0

  def check(self, st, where):
    got = {k: v.cpu().numpy() for k, v in st.items()}
    assert np.array_equal(got['ep_return'].view(np.int64), self.ret.view(np.int64)), where
    assert np.array_equal(got['ep_goal_touches'], self.touch), where
    assert np.array_equal(got['ep_len'], self.len), where
    assert np.array_equal(got['ep_count'], self.count), where
    seen = self.count > 0
    assert np.array_equal(got['last_return'][:, seen].view(np.int64), self.last_ret[:, seen].view(np.int64)), where
    assert np.array_equal(got['last_goal_touches'][seen], self.last_touch[seen]), where
    assert np.array_equal(got['last_len'][seen], self.last_len[seen]), where


@pytest.mark.parametrize('reward_dtype', ['f64', 'f32'])
def test_single_steps_and_fused_rollouts(reward_dtype):
  _need_gpu()
  import multigrid_amd
  n, A = 1500, 3
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=3, max_steps=40, reward_dtype=reward_dtype)
  st = env.enable_episode_stats(4.5)
  ref = RefStats(A, n, 4.5)
  env.reset(); ref.reset()
  for t in range(90):
    env.step_random()
    ref.step(env.reward.cpu().numpy(), env.done.cpu().numpy())
    if t % 15 == 14:
      ref.check(st, 'step %d' % t)
  for K in (32, 7):
    traj = env.rollout_random(K)
    r, d = traj['reward'].cpu().numpy(), traj['done'].cpu().numpy()
    for k in range(K):
      ref.step(r[k], d[k])
    ref.check(st, 'rollout of %d' % K)
  mask = np.arange(n) % 3 == 0                     # a partial reset drops those envs' episodes in progress
  env.reset(mask=mask); ref.reset(mask)
  env.step_random(); ref.step(env.reward.cpu().numpy(), env.done.cpu().numpy())
  ref.check(st, 'after masked reset')
  assert ref.count.sum() > 2 * n and ref.last_touch.max() > 0 and env.check_errors() == 0
  env.close()


def test_mutable_grid_kind_and_caller_reset():
  """DoorKey (the other kernel) with auto_reset off: done envs are reset by the caller's mask, as the reference's loop does."""
  _need_gpu()
  import torch
  import multigrid_amd
  n, A = 700, 2
  env = multigrid_amd.make_batched('MultiGrid-DoorKey-6x6-v0', n, seed=5, max_steps=25, auto_reset=False)
  A = env.n_agents
  st = env.enable_episode_stats(0.05)              # unshaped goal rewards: 1 - 0.9 * t / max_steps > 0.1
  ref = RefStats(A, n, 0.05)
  env.reset(); ref.reset()
  for t in range(80):
    env.step_random()
    d = env.done.cpu().numpy().astype(bool)
    ref.step(env.reward.cpu().numpy(), d)
    if d.any():
      env.reset(mask=torch.as_tensor(d, device=env.device)); ref.reset(d)
  ref.check(st, 'doorkey')
  assert ref.count.min() >= 3 and env.check_errors() == 0
  env.close()


def test_rollouts_of_a_mutable_grid_kind():
  """Rollouts on a mutable-grid handle run as one launch per step into the trajectory slots: the statistics follow every slot."""
  _need_gpu()
  import multigrid_amd
  n = 600
  env = multigrid_amd.make_batched('MultiGrid-DoorKey-6x6-v0', n, seed=2, max_steps=20, shaping='v8')
  st = env.enable_episode_stats(4.5)
  ref = RefStats(env.n_agents, n, 4.5)
  env.reset(); ref.reset()
  for K in (25, 9):
    traj = env.rollout_random(K)
    r, d = traj['reward'].cpu().numpy(), traj['done'].cpu().numpy()
    for k in range(K):
      ref.step(r[k], d[k])
    ref.check(st, 'doorkey rollout of %d' % K)
  assert ref.count.min() >= 1 and env.check_errors() == 0
  env.close()


def test_matches_the_trajectory_form_used_by_training():
  """ppo.EpisodeStats (the vectorised trajectory form, §8) and the library's per-step form agree on the same rollout."""
  _need_gpu()
  import multigrid_amd
  from multigrid_amd import ppo
  n = 1024
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=11, max_steps=50)
  st = env.enable_episode_stats(4.5)
  env.reset()
  traj = env.rollout_random(120)
  es = ppo.EpisodeStats(env.n_agents, n, env.device)
  es.add(traj['reward'], traj['done'])
  assert int(st['ep_count'].sum()) == int(es.totals[0]) > n
  # the trajectory form takes differences of running sums: equal up to float64 rounding, not bit for bit
  np.testing.assert_allclose(st['ep_return'].cpu().numpy(), es.ep_return.cpu().numpy(), rtol=0, atol=1e-9)
  assert np.array_equal(st['ep_goal_touches'].cpu().numpy(), es.ep_touches.cpu().numpy())
  env.close()
