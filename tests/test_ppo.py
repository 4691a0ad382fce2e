"""the batched v8 policy / GAE / PPO update against literal restatements of the
reference's per-sample code (v8_robust_ppo.py:35-102, 275-345).  Pure PyTorch, runs on CPU; the
env stand-in is the C oracle behind the BatchedMultiGrid tensor interface."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F  # noqa: E402

from multigrid_amd import ppo  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _literal_forward(sd, image_xyc, direction):
  """SimplePPOAgent.forward for ONE observation, written out op by op (v8:69-95):
  image [5,5,3] -> permute(2,0,1).unsqueeze(0) -> conv/relu x2 -> global average -> cat embed -> MLP."""
  x = image_xyc.permute(2, 0, 1).unsqueeze(0).float()
  x = F.relu(F.conv2d(x, sd['image_conv.0.weight'], sd['image_conv.0.bias'], padding=1))
  x = F.relu(F.conv2d(x, sd['image_conv.2.weight'], sd['image_conv.2.bias'], padding=1))
  x = x.mean(dim=(2, 3))
  e = sd['direction_embed.weight'][int(direction)].unsqueeze(0)
  h = torch.cat([x, e], dim=-1)
  h = F.relu(F.linear(h, sd['shared.0.weight'], sd['shared.0.bias']))
  h = F.relu(F.linear(h, sd['shared.2.weight'], sd['shared.2.bias']))
  return F.linear(h, sd['actor.weight'], sd['actor.bias'])[0], F.linear(h, sd['critic.weight'], sd['critic.bias'])[0, 0]


def test_agent_net_matches_reference_architecture():
  torch.manual_seed(0)
  net = ppo.AgentNet()
  sd = net.state_dict()
  # the reference's parameter names and shapes (v8:38-61), so its .pth checkpoints load as they are
  want = {'image_conv.0.weight': (32, 3, 3, 3), 'image_conv.0.bias': (32,), 'image_conv.2.weight': (64, 32, 3, 3),
          'image_conv.2.bias': (64,), 'direction_embed.weight': (4, 8), 'shared.0.weight': (128, 72),
          'shared.0.bias': (128,), 'shared.2.weight': (64, 128), 'shared.2.bias': (64,), 'actor.weight': (7, 64),
          'actor.bias': (7,), 'critic.weight': (1, 64), 'critic.bias': (1,)}
  assert {k: tuple(v.shape) for k, v in sd.items()} == want
  assert sum(v.numel() for v in sd.values()) == 37544                       # ~37.5 k params
  assert float(sd['actor.bias'].abs().sum()) == 0.0                           # zero biases (v8:67)
  w = sd['shared.2.weight']
  assert torch.allclose(w @ w.t(), 0.25 * torch.eye(64), atol=1e-5)            # orthogonal, gain 0.5 (v8:66)
  img = torch.randint(0, 11, (9, 5, 5, 3), dtype=torch.uint8)
  d = torch.randint(0, 4, (9,))
  logits, values = net(img, d)
  for i in range(9):
    l, v = _literal_forward(sd, img[i], d[i])
    assert torch.allclose(logits[i], l, atol=1e-6) and torch.allclose(values[i], v, atol=1e-6)


def _literal_gae(rewards, values, dones, gamma=0.99, lam=0.95):
  """compute_gae (v8:275-293) on Python lists."""
  adv, ret, a = [], [], 0
  for t in reversed(range(len(rewards))):
    nv = 0 if t == len(rewards) - 1 else values[t + 1]
    nt = 1 - dones[t]
    delta = rewards[t] + gamma * nv * nt - values[t]
    a = delta + gamma * lam * nt * a
    adv.insert(0, a)
    ret.insert(0, a + values[t])
  return adv, ret


def test_gae_matches_reference_loop():
  rng = np.random.RandomState(1)
  T, N = 37, 5
  r, v = rng.randn(T, N), rng.randn(T, N)
  dones = np.zeros(T, bool)
  dones[[11, 36]] = True
  adv, ret = ppo.compute_gae(torch.tensor(r), torch.tensor(v), torch.tensor(dones))
  for n in range(N):
    la, lr = _literal_gae(list(r[:, n]), list(v[:, n]), [float(x) for x in dones])
    assert np.allclose(adv[:, n].numpy(), la, atol=1e-12) and np.allclose(ret[:, n].numpy(), lr, atol=1e-12)


def test_ppo_loss_matches_reference_formula():
  torch.manual_seed(2)
  net = ppo.AgentNet()
  B = 33
  img = torch.randint(0, 11, (B, 5, 5, 3), dtype=torch.uint8)
  d = torch.randint(0, 4, (B,))
  act = torch.randint(0, 7, (B,))
  old_lp, adv, ret = -torch.rand(B) * 2, torch.randn(B), torch.randn(B)
  loss = ppo.ppo_loss(net, img, d, act, old_lp, adv, ret)
  logits, val = net(img, d)
  lp = torch.log_softmax(logits, -1)
  new_lp = lp.gather(1, act[:, None])[:, 0]
  ent = -(lp.exp() * lp).sum(-1)
  ratio = (new_lp - old_lp).exp()
  want = -torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv).mean() + 0.5 * ((val - ret) ** 2).mean() - 0.02 * ent.mean()
  assert torch.allclose(loss, want, atol=1e-6)


class OracleEnvTensors:
  """The C oracle behind BatchedMultiGrid's tensor interface (CPU stand-in for the device env)."""

  def __init__(self, n, seed=0, gid_base=0):
    from multigrid_amd import layouts
    from oracle import c_oracle
    self.o = c_oracle.OracleBatch(layouts.cluttered_fixed_15x15(), n, shaping=1, auto_reset=1, seed=seed, gid_base=gid_base)
    self.n_envs, self.n_agents, self.stride = n, 3, self.o.stride
    self.actions = torch.full((self.stride, 4), 6, dtype=torch.uint8)
    self.o.reset()
    self._sync()

  def _sync(self):
    n = self.n_envs
    self.obs_image = torch.from_numpy(self.o.obs_image[:, :n].reshape(3, n, 5, 5, 3).copy())
    self.obs_dir = torch.from_numpy(self.o.obs_dir[:, :n].copy())
    self.reward = torch.from_numpy(self.o.reward[:, :n].copy())
    self.done = torch.from_numpy(self.o.done[:n].copy())

  def step(self):
    self.o.step(np.ascontiguousarray(self.actions.numpy()[:, :3].T))
    self._sync()


def test_rollout_and_update_run_on_the_env_interface():
  env = OracleEnvTensors(48, seed=3)
  ctl = ppo.BatchedController(env, device=torch.device('cpu'), ddp=False, seed=5)
  traj = ctl.rollout(101)
  assert traj['image'][0].shape == (101, 48, 5, 5, 3) and traj['done'].shape == (101, 48)
  assert bool(traj['done'][99].all()) and not bool(traj['done'][:99].any())      # episodes end at step 100
  assert int(traj['action'][2].max()) <= 6
  before = [p.detach().clone() for p in ctl.nets[0].parameters()]
  losses = ctl.update(traj)
  assert len(losses) == 3 and all(np.isfinite(losses))
  assert any(not torch.equal(a, b) for a, b in zip(before, ctl.nets[0].parameters()))


def _ddp_worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from multigrid_amd import ppo as P, shard
  from test_ppo import OracleEnvTensors
  base, n = shard.shard_range(rank, world, 16)
  env = OracleEnvTensors(n, seed=3, gid_base=base)          # different envs per rank
  ctl = P.BatchedController(env, device=torch.device('cpu'), seed=5)
  assert ctl.ddp
  torch.manual_seed(100 + rank)                             # different action samples per rank
  traj = ctl.rollout(40)
  ctl.update(traj)
  torch.save([p.detach() for p in ctl.nets[1].parameters()], os.path.join(out_dir, 'p%d.pt' % rank))
  # episode statistics: every rank's own sums, then the all-reduced total must be the same on both
  ctl.stats.add(torch.stack(traj['reward'], dim=1), traj['done'])
  local = ctl.stats.totals.clone()
  red = ctl.stats.reduce()
  torch.save({'local': local, 'reduced': red}, os.path.join(out_dir, 's%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


def test_ddp_gradient_allreduce_keeps_ranks_in_lockstep(tmp_path):
  """(also: EpisodeStats.reduce() sums the ranks' finished-episode totals)"""
  import torch.multiprocessing as mp
  port = 29700 + (os.getpid() % 1500)
  mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  p0 = torch.load(os.path.join(str(tmp_path), 'p0.pt'), weights_only=True)
  p1 = torch.load(os.path.join(str(tmp_path), 'p1.pt'), weights_only=True)
  assert all(torch.equal(a, b) for a, b in zip(p0, p1))     # same weights after 4 all-reduced epochs
  s0 = torch.load(os.path.join(str(tmp_path), 's0.pt'), weights_only=True)
  s1 = torch.load(os.path.join(str(tmp_path), 's1.pt'), weights_only=True)
  assert s0['reduced'] == s1['reduced']
  assert s0['reduced']['episodes'] == int(s0['local'][0] + s1['local'][0])
  assert s0['reduced']['goal_touches'] == int(s0['local'][2] + s1['local'][2])


def test_multigrid_net_matches_reference_layout():
  torch.manual_seed(1)
  net = ppo.MultiGridNet(n_actions=7, n_agents=3)
  sd = net.state_dict()
  assert tuple(sd['image_layers.0.weight'].shape) == (32, 3, 3, 3) and tuple(sd['image_layers.5.weight'].shape) == (64, 64)
  assert tuple(sd['direction_layers.0.weight'].shape) == (8, 12) and tuple(sd['head.0.weight'].shape) == (192, 72)
  img = torch.randint(0, 11, (4, 5, 5, 3), dtype=torch.uint8)
  dirs = torch.randint(0, 4, (4, 3))
  q = net(img, dirs)
  # literal forward for sample 2 (multigrid_network.py:57-79)
  x = img[2:3].permute(0, 3, 1, 2).float()
  x = F.leaky_relu(F.conv2d(x, sd['image_layers.0.weight'], sd['image_layers.0.bias']))
  x = F.leaky_relu(F.conv2d(x, sd['image_layers.2.weight'], sd['image_layers.2.bias'])).reshape(1, -1)
  x = F.leaky_relu(F.linear(x, sd['image_layers.5.weight'], sd['image_layers.5.bias']))
  oh = F.one_hot(dirs[2:3], 4).reshape(1, -1).float()
  e = F.relu(F.linear(oh, sd['direction_layers.0.weight'], sd['direction_layers.0.bias']))
  h = torch.cat([x, e], -1)
  h = F.relu(F.linear(h, sd['head.0.weight'], sd['head.0.bias']))
  h = F.relu(F.linear(h, sd['head.2.weight'], sd['head.2.bias']))
  want = F.linear(h, sd['head.4.weight'], sd['head.4.bias'])
  assert q.shape == (4, 7) and torch.allclose(q[2:3], want, atol=1e-6)


@pytest.mark.gpu
def test_gpu_rollout_update_on_device_env():
  """Config 5 shape on one GPU: HIP env + three batched nets, zero-copy obs, no host sync per step."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', 2048, seed=1)
  env.reset()
  ctl = ppo.BatchedController(env, ddp=False, seed=2)
  traj = ctl.rollout(100)
  assert traj['image'][0].is_cuda and traj['image'][0].shape == (100, 2048, 5, 5, 3)
  assert bool(traj['done'][99].all().item()) and not bool(traj['done'][:99].any().item())
  assert int(torch.stack(traj['action']).max().item()) <= 6
  losses = ctl.update(traj)
  assert all(np.isfinite(losses)) and env.check_errors() == 0
  # the stored observation of step t is what the env held before step t: re-derive step 0's obs
  env2 = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', 2048, seed=1)
  env2.reset()
  assert torch.equal(traj['image'][1][0], env2.obs_image[1])


def test_episode_stats_match_a_per_env_loop():
  """EpisodeStats against the reference's per-episode bookkeeping (v8_robust_ppo.py:233-271, 425) done env by env."""
  import torch
  rng = np.random.RandomState(0)
  A, N, T = 3, 37, 120
  reward = rng.choice([0.0, 0.03, -0.05, 5.0], size=(T, A, N), p=[.5, .3, .15, .05])
  done = rng.rand(T, N) < 0.04
  st = ppo.EpisodeStats(A, N, torch.device('cpu'))
  st.add(torch.from_numpy(reward), torch.from_numpy(done))
  episodes, total, touches = 0, 0.0, 0
  for n in range(N):
    ep, tc = np.zeros(A), 0
    for t in range(T):
      ep += reward[t, :, n]
      tc += int((reward[t, :, n] >= 4.5).sum())
      if done[t, n]:
        episodes += 1; total += ep.sum(); touches += tc
        ep, tc = np.zeros(A), 0
  out = st.reduce()
  assert out['episodes'] == episodes and out['goal_touches'] == touches
  assert abs(out['mean_collective_return'] - total / episodes) < 1e-9


def test_save_and_load_models_round_trip(tmp_path):
  import torch

  class FakeEnv:
    n_agents, n_envs = 2, 4
    obs_image = torch.zeros((2, 4, 5, 5, 3), dtype=torch.uint8)
  a = ppo.BatchedController(FakeEnv(), ddp=False, seed=1)
  b = ppo.BatchedController(FakeEnv(), ddp=False, seed=2)
  a.save_models(str(tmp_path / 'm' / 'best'))
  assert (tmp_path / 'm' / 'best_agent_1.pth').exists()
  b.load_models(str(tmp_path / 'm' / 'best'))
  for x, y in zip(a.nets, b.nets):
    for p, q in zip(x.parameters(), y.parameters()):
      assert torch.equal(p, q)


@pytest.mark.gpu
def test_graph_rollout_is_self_consistent():
  """The HIP-graph rollout stores exactly what happened: replaying its stored actions through a second env with
  the same seed reproduces its stored obs / rewards / done, and the nets recompute its stored log-probs and values."""
  torch = pytest.importorskip('torch')
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  from multigrid_amd import ppo
  n, T = 512, 40
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=21)
  env.reset()
  ctl = ppo.BatchedController(env, ddp=False, seed=3)
  traj = ctl.rollout(T, use_graph=True)
  twin = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=21)
  twin.reset()
  for t in range(T):
    for a in range(3):
      assert torch.equal(traj['image'][a][t], twin.obs_image[a]), (t, a)
      assert torch.equal(traj['direction'][a][t], twin.obs_dir[a]), (t, a)
      _, logp, _, val = ctl.nets[a].get_action_and_value(twin.obs_image[a], twin.obs_dir[a], traj['action'][a][t])
      assert torch.allclose(logp, traj['log_prob'][a][t], atol=1e-5) and torch.allclose(val, traj['value'][a][t], atol=1e-5)
    acts = torch.stack([traj['action'][a][t] for a in range(3)])
    assert int(acts.min()) >= 0 and int(acts.max()) <= 6
    twin.step(acts)
    for a in range(3):
      assert torch.equal(traj['reward'][a][t], twin.reward[a].float()), (t, a)
    assert torch.equal(traj['done'][t], twin.done), t
  # a second graphed rollout continues from where the first stopped, and an update runs on its output
  traj = ctl.rollout(T, use_graph=True)
  losses = ctl.update(traj, epochs=1)
  assert all(np.isfinite(l) for l in losses)
  env.close(); twin.close()


@pytest.mark.gpu
def test_untrained_returns_agree_with_the_reference_log():
  """Statistical sanity only, not parity: the reference's own v8 run (models8/results_simple_...json, `collective_rewards`)
  starts at -4.58, -8.11, -9.38, -9.23, -7.98 per episode (mean of its first ten: -6.3, s.d. 3.6) before learning takes
  hold.  The whole path here (layout, spawn, step, v8 shaping, freshly initialised nets, 100-step episodes) must land
  in the same place: 4096 untrained episodes average about -7.0."""
  torch = pytest.importorskip('torch')
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  from multigrid_amd import ppo
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', 4096, seed=5)
  env.reset()
  ctl = ppo.BatchedController(env, ddp=False, seed=6)
  traj = ctl.rollout(100)
  ret = torch.stack(traj['reward']).double().sum(dim=(0, 1))          # collective return per episode (v8:425)
  mean, sd = float(ret.mean()), float(ret.std())
  assert -9.5 < mean < -4.5, mean          # reference's first ten episodes: -6.3 +- 3.6 / sqrt(10)
  assert 1.5 < sd < 8.0, sd                # its episode-to-episode spread there: 3.6
