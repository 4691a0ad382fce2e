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
  # [T, A, N] rewards with per-env done flags [T, N] (what BatchedController.advantages passes), odd and tiny T
  for T in (1, 2, 3, 100, 129):
    A = 3
    r, v = rng.randn(T, A, N).astype(np.float32), rng.randn(T, A, N).astype(np.float32)
    dn = rng.rand(T, N) < 0.1
    adv, ret = ppo.compute_gae(torch.tensor(r), torch.tensor(v), torch.tensor(dn))
    assert adv.dtype == torch.float64 and adv.shape == (T, A, N)
    for a in range(A):
      for n in range(N):
        la, lr = _literal_gae([float(x) for x in r[:, a, n]], [float(x) for x in v[:, a, n]], [float(x) for x in dn[:, n]])
        assert np.allclose(adv[:, a, n].numpy(), la, atol=1e-12) and np.allclose(ret[:, a, n].numpy(), lr, atol=1e-12)


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
  assert traj['image'].shape == (101, 3, 48, 5, 5, 3) and traj['done'].shape == (101, 48)
  assert bool(traj['done'][99].all()) and not bool(traj['done'][:99].any())      # episodes end at step 100
  assert int(traj['action'].max()) <= 6
  before = [p.detach().clone() for p in ctl.nets[0].parameters()]
  losses = ctl.update(traj)
  assert len(losses) == 3 and all(np.isfinite(losses))
  assert any(not torch.equal(a, b) for a, b in zip(before, ctl.nets[0].parameters()))


def _reference_update_agent(net, opt, images, dirs, acts, old_lp, rewards, values, dones, epochs=4):
  """update_agent (v8_robust_ppo.py:295-345) for ONE agent and ONE env's episode, as the reference writes it: GAE on
  Python lists, FloatTensor casts, normalisation when std > 0, 4 full-batch epochs, clip_grad_norm_ 0.5."""
  adv, ret = _literal_gae([float(x) for x in rewards], [float(x) for x in values], [float(x) for x in dones])
  adv, ret = torch.FloatTensor(adv), torch.FloatTensor(ret)
  if adv.std() > 0:
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
  losses = []
  for _ in range(epochs):
    _, new_lp, ent, new_v = net.get_action_and_value(images, dirs, acts)
    ratio = torch.exp(new_lp - old_lp)
    policy_loss = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
    value_loss = 0.5 * (new_v - ret).pow(2).mean()
    loss = policy_loss + value_loss + -0.02 * ent.mean()
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
    opt.step()
    losses.append(float(loss.detach()))
  return losses


@pytest.mark.parametrize('conv_mode', ['grouped', 'loop'])
def test_stacked_update_is_the_references_update_agent_per_agent(conv_mode):
  """The stacked controller (one forward / backward for all agents, one Adam, per-agent clipping) against the
  reference's update_agent run agent by agent on the same trajectory of ONE env: same losses, same weights after."""
  env = OracleEnvTensors(1, seed=4)
  ctl = ppo.BatchedController(env, device=torch.device('cpu'), ddp=False, seed=9, conv_mode=conv_mode)
  refs = ctl.nets                                            # stand-alone copies of the initial weights
  opts = [torch.optim.Adam(n.parameters(), lr=1e-4) for n in refs]
  traj = ctl.rollout(100, use_graph=False)
  want = []
  for a in range(3):
    want.append(_reference_update_agent(refs[a], opts[a], traj['image'][:, a, 0], traj['direction'][:, a, 0], traj['action'][:, a, 0],
                                        traj['log_prob'][:, a, 0], traj['reward'][:, a, 0], traj['value'][:, a, 0], traj['done'][:, 0]))
  got = ctl.update(traj)
  for a in range(3):
    assert abs(got[a] - np.mean(want[a])) < 2e-5, (a, got[a], want[a])
    for p, q in zip(ctl.nets[a].parameters(), refs[a].parameters()):
      assert torch.allclose(p, q, atol=2e-6), a


def test_stacked_forward_equals_the_per_agent_nets():
  torch.manual_seed(3)
  st = ppo.StackedAgentNets(3)
  img = torch.randint(0, 11, (3, 17, 5, 5, 3), dtype=torch.uint8)
  d = torch.randint(0, 4, (3, 17))
  logits, values = st(img, d)
  st2 = ppo.StackedAgentNets(3, conv_mode='loop')
  st2.load_state_dict(st.state_dict())
  l2, v2 = st2(img, d)
  assert torch.allclose(logits, l2, atol=1e-6) and torch.allclose(values, v2, atol=1e-6)
  for a in range(3):
    l, v = st.agent_net(a)(img[a], d[a])
    assert torch.allclose(logits[a], l, atol=1e-6) and torch.allclose(values[a], v, atol=1e-6)
  # per-agent gradient clipping = clip_grad_norm_ on each agent's own parameters
  (logits.sum() * 3.0 + values.pow(2).sum()).backward()
  nets = [st.agent_net(a) for a in range(3)]
  for a, net in enumerate(nets):
    l, v = net(img[a], d[a])
    (l.sum() * 3.0 + v.pow(2).sum()).backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
  st.clip_grad_norm_(0.5)
  for a, net in enumerate(nets):
    sd = dict(net.named_parameters())
    for ref_name, name, _ in ppo._STACK:
      assert torch.allclose(getattr(st, name).grad[a], sd[ref_name].grad, atol=1e-6), (a, ref_name)


def test_multigrid_net_policy_rolls_out_and_updates():
  """BASELINE config 5's policy: networks/multigrid_network.py per agent, fed EVERY agent's direction."""
  env = OracleEnvTensors(24, seed=6)
  ctl = ppo.BatchedController(env, device=torch.device('cpu'), ddp=False, seed=5, policy='multigrid_net')
  net = ctl.policy.nets[1]
  assert isinstance(net, ppo.MultiGridNet) and tuple(net.direction_layers[0].weight.shape) == (8, 12)
  traj = ctl.rollout(30, use_graph=False)
  # the stored log-probs / values are what agent 1's net computes from its image and ALL agents' directions
  t = 7
  logits, val = net(traj['image'][t, 1], traj['direction'][t].t())
  lp = torch.log_softmax(logits, -1).gather(1, traj['action'][t, 1][:, None])[:, 0]
  assert torch.allclose(lp, traj['log_prob'][t, 1], atol=1e-6) and torch.allclose(val, traj['value'][t, 1], atol=1e-6)
  # ... and they do depend on another agent's direction
  other = traj['direction'][t].t().clone()
  other[:, 2] = (other[:, 2] + 1) % 4
  assert not torch.allclose(net(traj['image'][t, 1], other)[0], logits)
  before = [p.detach().clone() for p in net.parameters()]
  losses = ctl.update(traj)
  assert len(losses) == 3 and all(np.isfinite(losses))
  assert any(not torch.equal(a, b) for a, b in zip(before, net.parameters()))
  # a reference MultiGridNetwork state_dict (no critic) loads into agent 0
  ref_sd = ppo.MultiGridNet(n_agents=3).state_dict()
  ctl.policy.load_agent_state_dict(0, ref_sd)
  assert torch.equal(ctl.policy.nets[0].head[4].weight, ref_sd['head.4.weight'])


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
  traj = ctl.rollout(40)
  ctl.update(traj)
  torch.save([p.detach() for p in ctl.nets[1].parameters()], os.path.join(out_dir, 'p%d.pt' % rank))
  torch.save(traj['action'][:5].clone(), os.path.join(out_dir, 'a%d.pt' % rank))
  # episode statistics: every rank's own sums, then the all-reduced total must be the same on both
  ctl.stats.add(traj['reward'], traj['done'])
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
  # the controller re-seeds its sampler per rank: same weights, same seed argument, different action noise
  a0 = torch.load(os.path.join(str(tmp_path), 'a0.pt'), weights_only=True)
  a1 = torch.load(os.path.join(str(tmp_path), 'a1.pt'), weights_only=True)
  assert not torch.equal(a0, a1)
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
  assert traj['image'].is_cuda and traj['image'].shape == (100, 3, 2048, 5, 5, 3)
  assert bool(traj['done'][99].all().item()) and not bool(traj['done'][:99].any().item())
  assert int(traj['action'].max().item()) <= 6
  losses = ctl.update(traj)
  assert all(np.isfinite(losses)) and env.check_errors() == 0
  # the stored observation of step t is what the env held before step t: re-derive step 0's obs
  env2 = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', 2048, seed=1)
  env2.reset()
  assert torch.equal(traj['image'][0, 1], env2.obs_image[1])


@pytest.mark.gpu
def test_two_rank_training_on_one_gpu_keeps_the_ranks_in_lockstep(tmp_path):
  """BASELINE config 5's multi-rank leg, rehearsed the only way one card allows: two ranks of `python -m
  multigrid_amd.train` under torch.distributed.run, gloo, both on cuda:0 — sharded envs, HIP-graph rollouts, four
  epochs per iteration with one flat gradient all-reduce each, statistics all-reduced.  Round 2's rehearsal stood still
  inside loss.backward() (DistributedDataParallel's reducer driving gloo on device buckets); the flat all-reduce does
  not.  Done = three iterations printed and bit-identical weights on both ranks."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import json
  import subprocess
  port = 29800 + (os.getpid() % 150)
  prefix = os.path.join(str(tmp_path), 'w')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(port), '-m', 'multigrid_amd.train', '--backend', 'gloo', '--all-on-device0', '--envs', '512',
         '--steps', '25', '--iters', '3', '--hang-dump', '60', '--dump-weights', prefix, '--out', os.path.join(str(tmp_path), 'best')]
  r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=200)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
  assert [l['iter'] for l in lines] == [0, 1, 2] and all(np.isfinite(l['loss']).all() for l in lines)
  w0 = torch.load(prefix + '.rank0.pt', weights_only=True)
  w1 = torch.load(prefix + '.rank1.pt', weights_only=True)
  assert sorted(w0) == sorted(w1) and all(torch.equal(w0[k], w1[k]) for k in w0)


@pytest.mark.gpu
@pytest.mark.parametrize('policy', ['simple', 'multigrid_net'])
def test_gpu_gae_and_loss_against_the_reference_loops(policy):
  """On the device, on a real rollout of the HIP env: the scan GAE + per-agent normalisation against the reference's
  list loop (v8:275-293, 308-317) for sampled (agent, env) pairs, and the stacked loss against the formula evaluated
  agent by agent with the stand-alone nets (v8:319-336)."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import multigrid_amd
  n, T = 256, 100
  env = multigrid_amd.make_batched('MultiGrid-Cluttered-Fixed-15x15', n, seed=8)
  env.reset()
  ctl = ppo.BatchedController(env, ddp=False, seed=4, policy=policy)
  traj = ctl.rollout(T)
  adv_raw, ret_raw = ppo.compute_gae(traj['reward'], traj['value'], traj['done'])
  rw, vl, dn = traj['reward'].cpu().numpy(), traj['value'].cpu().numpy(), traj['done'].cpu().numpy()
  for a, e in [(0, 0), (1, 17), (2, 255), (1, 100)]:
    la, lr = _literal_gae([float(x) for x in rw[:, a, e]], [float(x) for x in vl[:, a, e]], [float(x) for x in dn[:, e]])
    assert np.allclose(adv_raw[:, a, e].cpu().numpy(), la, atol=1e-9) and np.allclose(ret_raw[:, a, e].cpu().numpy(), lr, atol=1e-9)
  adv, ret = ctl.advantages(traj)
  with torch.no_grad():
    logits, values = ctl.policy(traj['image'].transpose(0, 1).flatten(1, 2), traj['direction'].transpose(0, 1).flatten(1, 2))
    got = ppo.ppo_loss_terms(logits, values, traj['action'].transpose(0, 1).flatten(1, 2),
                             traj['log_prob'].transpose(0, 1).flatten(1, 2), adv, ret)
    for a in range(3):
      # reference normalisation of this agent's advantages (v8:308-317), all (t, env) samples of the agent as one batch
      fa = torch.FloatTensor(adv_raw[:, a].cpu().numpy().reshape(-1)).to(env.device)
      fa = (fa - fa.mean()) / (fa.std() + 1e-8)
      fr = ret_raw[:, a].float().reshape(-1)
      assert torch.allclose(adv[a], fa, atol=1e-5) and torch.allclose(ret[a], fr, atol=1e-6)
      net = ctl.nets[a]
      img, d = traj['image'][:, a].flatten(0, 1), traj['direction'][:, a].flatten(0, 1)
      if policy == 'multigrid_net':
        lg, v = net(img, traj['direction'].permute(0, 2, 1).flatten(0, 1))
      else:
        lg, v = net(img, d)
      lp_all = torch.log_softmax(lg, -1)
      lp = lp_all.gather(1, traj['action'][:, a].flatten(0, 1)[:, None])[:, 0]
      ratio = (lp - traj['log_prob'][:, a].flatten(0, 1)).exp()
      want = (-torch.min(ratio * fa, ratio.clamp(0.8, 1.2) * fa).mean() + 0.5 * ((v - fr) ** 2).mean()
              - 0.02 * (-(lp_all.exp() * lp_all).sum(-1)).mean())
      assert abs(float(got[a]) - float(want)) < 1e-4, (a, float(got[a]), float(want))
  losses = ctl.update(traj)
  assert all(np.isfinite(losses)) and env.check_errors() == 0
  env.close()


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
  # the files hold the reference's keys and shapes (v8:362-366): SimplePPOAgent.load_state_dict would take them
  sd = torch.load(str(tmp_path / 'm' / 'best_agent_0.pth'), weights_only=True)
  assert set(sd) == set(ppo.AgentNet().state_dict()) and tuple(sd['critic.weight'].shape) == (1, 64)


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
  nets = ctl.nets
  for t in range(T):
    for a in range(3):
      assert torch.equal(traj['image'][t, a], twin.obs_image[a]), (t, a)
      assert torch.equal(traj['direction'][t, a], twin.obs_dir[a]), (t, a)
      _, logp, _, val = nets[a].get_action_and_value(twin.obs_image[a], twin.obs_dir[a], traj['action'][t, a])
      assert torch.allclose(logp, traj['log_prob'][t, a], atol=1e-5) and torch.allclose(val, traj['value'][t, a], atol=1e-5)
    acts = traj['action'][t]
    assert int(acts.min()) >= 0 and int(acts.max()) <= 6
    twin.step(acts)
    assert traj['reward'].dtype == torch.float64 and torch.equal(traj['reward'][t], twin.reward), t   # float64: GAE runs on them uncast (v8:275-293)
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
  ret = traj['reward'].double().sum(dim=(0, 1))                        # collective return per episode (v8:425)
  mean, sd = float(ret.mean()), float(ret.std())
  assert -9.5 < mean < -4.5, mean          # reference's first ten episodes: -6.3 +- 3.6 / sqrt(10)
  assert 1.5 < sd < 8.0, sd                # its episode-to-episode spread there: 3.6

