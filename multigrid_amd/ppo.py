"""Batched v8 PPO on top of the device-resident env.

PyTorch-ROCm only (no HIP kernels here): the policy consumes `env.obs_image[a]` / `env.obs_dir[a]`
in place, writes its sampled actions into `env.actions[:, a]`, and `env.step()` is one kernel
launch on the same stream — no host round trip per step, where the reference does 3 batch-1
forwards with an `.item()` sync each (v8_robust_ppo.py:138-157).

Reference anchors (relative to /root/reference):
  v8_robust_ppo.py:35-102   SimplePPOAgent (architecture, init, get_action_and_value)
  v8_robust_ppo.py:275-293  compute_gae   (gamma .99, lambda .95; next_value = 0 at the last step)
  v8_robust_ppo.py:295-345  update_agent  (4 full-batch epochs, clip [0.8, 1.2], vf .5, ent .02,
                                           grad-clip .5, advantages normalised when std > 0)
  v8_robust_ppo.py:117-121  one independent net + Adam(lr=1e-4) per agent
Module names match the reference's, so a reference `state_dict` checkpoint
(v8_robust_ppo.py:362-366) loads with `load_state_dict(torch.load(path, weights_only=True))`.
"""
import torch
from torch import nn
from torch.distributions import Categorical


class AgentNet(nn.Module):
  """SimplePPOAgent (v8_robust_ppo.py:35-102), batch-first only."""

  def __init__(self, n_actions=7):
    super().__init__()
    self.image_conv = nn.Sequential(
        nn.Conv2d(3, 32, 3, padding=1), nn.ReLU(),
        nn.Conv2d(32, 64, 3, padding=1), nn.ReLU(),
        nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())
    self.direction_embed = nn.Embedding(4, 8)
    self.shared = nn.Sequential(nn.Linear(72, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU())
    self.actor = nn.Linear(64, n_actions)
    self.critic = nn.Linear(64, 1)
    for m in self.modules():                       # v8:63-67
      if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight, 0.5)
        m.bias.data.zero_()

  def forward(self, image, direction):
    """image: uint8/float [B, V, V, 3] laid out [x][y][c] as the env writes it; direction: [B]."""
    x = image.permute(0, 3, 1, 2).float()          # v8:75 — channels first, x/y kept as they are
    feat = torch.cat([self.image_conv(x), self.direction_embed(direction.long())], dim=-1)
    h = self.shared(feat)
    return self.actor(h), self.critic(h).squeeze(-1)

  def get_action_and_value(self, image, direction, action=None):
    logits, value = self.forward(image, direction)
    dist = Categorical(logits=logits, validate_args=False)   # argument validation synchronises with the host
    if action is None:
      action = dist.sample()
    return action, dist.log_prob(action), dist.entropy(), value


class MultiGridNet(nn.Module):
  """networks/multigrid_network.py:8-79 (MultiGridNetwork), batch-first: two VALID 3x3 convs
  (5x5 -> 1x1), one-hot directions of ALL agents -> fc_direction, head 192 -> 64 -> n_actions.
  kernel_size / fc_direction default to config/default.yaml (architecture block).  Same module names as the
  reference, so its state_dicts load."""

  def __init__(self, n_actions=7, n_agents=3, kernel_size=3, fc_direction=8):
    super().__init__()
    self.n_agents = n_agents
    self.image_layers = nn.Sequential(
        nn.Conv2d(3, 32, (kernel_size, kernel_size)), nn.LeakyReLU(),
        nn.Conv2d(32, 64, (kernel_size, kernel_size)), nn.LeakyReLU(),
        nn.Flatten(), nn.Linear(64, 64), nn.LeakyReLU())
    self.direction_layers = nn.Sequential(nn.Linear(4 * n_agents, fc_direction), nn.ReLU())
    self.head = nn.Sequential(nn.Linear(64 + fc_direction, 192), nn.ReLU(), nn.Linear(192, 64), nn.ReLU(),
                              nn.Linear(64, n_actions))

  def forward(self, image, directions):
    """image [B, V, V, 3]; directions [B, n_agents] (every agent's compass, multigrid_network.py:70-74)."""
    x = self.image_layers(image.permute(0, 3, 1, 2).float())
    onehot = torch.nn.functional.one_hot(directions.long(), num_classes=4).reshape(image.shape[0], -1).float()
    return self.head(torch.cat([x, self.direction_layers(onehot)], dim=-1))


def compute_gae(rewards, values, dones, gamma=0.99, gae_lambda=0.95):
  """compute_gae (v8_robust_ppo.py:275-293) vectorised over the env axis.

  rewards, values: float [T, N]; dones: bool/0-1 [T] or [T, N].  Returns (advantages, returns).
  Quirks kept: the value after the last stored step is 0 (no bootstrap), and `done` of step t
  gates the bootstrap from step t+1.
  """
  T = rewards.shape[0]
  dones = dones.to(rewards.dtype)
  if dones.dim() == 1:
    dones = dones.unsqueeze(-1).expand_as(rewards)
  adv = torch.zeros_like(rewards)
  running = torch.zeros_like(rewards[0])
  for t in reversed(range(T)):
    next_value = torch.zeros_like(values[0]) if t == T - 1 else values[t + 1]
    nonterm = 1.0 - dones[t]
    delta = rewards[t] + gamma * next_value * nonterm - values[t]
    running = delta + gamma * gae_lambda * nonterm * running
    adv[t] = running
  return adv, adv + values


def ppo_loss(model, images, directions, actions, old_log_probs, advantages, returns,
             clip_low=0.8, clip_high=1.2, vf_coef=0.5, ent_coef=0.02):
  """The loss of one epoch of update_agent (v8_robust_ppo.py:326-336).  `model(images, directions)`
  -> (logits, values): an AgentNet or its DistributedDataParallel wrapper (DDP hooks `forward`)."""
  logits, new_values = model(images, directions)
  dist = Categorical(logits=logits, validate_args=False)   # argument validation synchronises with the host
  new_log_probs, entropy = dist.log_prob(actions), dist.entropy()
  ratio = torch.exp(new_log_probs - old_log_probs)
  surr1 = ratio * advantages
  surr2 = torch.clamp(ratio, clip_low, clip_high) * advantages
  policy_loss = -torch.min(surr1, surr2).mean()
  value_loss = vf_coef * (new_values - returns).pow(2).mean()
  entropy_loss = -ent_coef * entropy.mean()
  return policy_loss + value_loss + entropy_loss


class EpisodeStats:
  """run_episode's bookkeeping (v8_robust_ppo.py:233-271) for N envs at once, on device: per-agent return of the
  episode in progress, goal touches (shaped reward >= 4.5, v8:246-248), and sums over finished episodes.
  `reduce()` adds the finished-episode sums over all ranks (one small all-reduce: RCCL on GPUs, gloo in tests)."""

  def __init__(self, n_agents, n_envs, device):
    self.ep_return = torch.zeros((n_agents, n_envs), dtype=torch.float64, device=device)
    self.ep_touches = torch.zeros(n_envs, dtype=torch.int64, device=device)
    # finished episodes: [count, sum of collective returns, sum of goal touches]
    self.totals = torch.zeros(3, dtype=torch.float64, device=device)

  @torch.no_grad()
  def add(self, reward, done):
    """reward [A, N] (or [T, A, N]), done [N] (or [T, N]) of one step (or T steps in order)."""
    if reward.dim() == 3:
      for t in range(reward.shape[0]):
        self.add(reward[t], done[t])
      return
    self.ep_return += reward.to(torch.float64)
    self.ep_touches += (reward >= 4.5).sum(dim=0)
    fin = done.bool()
    self.totals[0] += fin.sum()
    self.totals[1] += self.ep_return.sum(dim=0)[fin].sum()       # collective reward = sum over agents (v8:425)
    self.totals[2] += self.ep_touches[fin].sum()
    self.ep_return[:, fin] = 0
    self.ep_touches[fin] = 0

  def reduce(self):
    """{'episodes', 'mean_collective_return', 'goal_touches'} over every rank's finished episodes."""
    import torch.distributed as dist
    t = self.totals.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
      dist.all_reduce(t)
    n = float(t[0])
    return {'episodes': int(n), 'mean_collective_return': float(t[1]) / n if n else 0.0, 'goal_touches': int(t[2])}


class BatchedController:
  """SimpleController (v8_robust_ppo.py:105-366) over N envs at once.

  `env` is a BatchedMultiGrid (or anything with its tensors: obs_image [A,N,V,V,3], obs_dir
  [A,N], reward [A,N], done [N], actions [stride,4], n_envs, n_agents, step(), reset()).
  With torch.distributed initialised, every net is wrapped in DistributedDataParallel so the
  per-agent gradients are all-reduced (RCCL over xGMI on GPUs, gloo in the CPU tests); envs stay
  sharded and nothing else crosses ranks.
  """

  def __init__(self, env, lr=1e-4, device=None, ddp=None, seed=0):
    self.env = env
    self.n_agents = env.n_agents
    self.device = device if device is not None else env.obs_image.device
    torch.manual_seed(seed)                       # same initial weights on every rank
    self.nets = [AgentNet().to(self.device) for _ in range(self.n_agents)]
    import torch.distributed as dist
    if ddp is None:
      ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    self.ddp = ddp
    if ddp:
      from torch.nn.parallel import DistributedDataParallel as DDP
      ids = [self.device.index] if self.device.type == 'cuda' else None
      self.models = [DDP(n, device_ids=ids) for n in self.nets]
    else:
      self.models = self.nets
    self.optimizers = [torch.optim.Adam(n.parameters(), lr=lr) for n in self.nets]
    self.stats = EpisodeStats(self.n_agents, env.n_envs, self.device)

  def save_models(self, path_prefix):
    """save_models (v8_robust_ppo.py:362-366): one state_dict per agent, the reference's file names."""
    import os
    os.makedirs(os.path.dirname(path_prefix) or '.', exist_ok=True)
    for i, net in enumerate(self.nets):
      torch.save(net.state_dict(), '%s_agent_%d.pth' % (path_prefix, i))

  def load_models(self, path_prefix):
    """Loads files written by save_models — or by the reference: the module names are the same."""
    for i, net in enumerate(self.nets):
      net.load_state_dict(torch.load('%s_agent_%d.pth' % (path_prefix, i), map_location=self.device, weights_only=True))

  @torch.no_grad()
  def rollout(self, n_steps, use_graph=None):
    """run_episode (v8_robust_ppo.py:222-273) for N envs x n_steps, everything on device.
    Returns per-agent lists of [T, N, ...] tensors.

    use_graph (default: on for CUDA devices): the per-step policy work — A forwards, sampling, the action
    write and the trajectory stores — is launch-bound at these sizes (≈ 60 tiny kernels per step), so it
    is captured once into two HIP graphs and replayed around the env's own kernel launch: a step is then
    graph replay + `mg_step` + graph replay.  The returned tensors are the graph's own buffers: they are
    overwritten by the next graphed rollout of the same length."""
    if use_graph is None:
      use_graph = self.device.type == 'cuda'
    if use_graph:
      return self._graph_rollout(n_steps)
    env, A, N = self.env, self.n_agents, self.env.n_envs
    buf = {k: [[] for _ in range(A)] for k in ('image', 'direction', 'action', 'log_prob', 'value', 'reward')}
    dones = []
    for _ in range(n_steps):
      for a in range(A):
        img, d = env.obs_image[a].clone(), env.obs_dir[a].clone()
        act, logp, _, val = self.nets[a].get_action_and_value(img, d)
        env.actions[:N, a] = act.to(torch.uint8)
        for k, v in (('image', img), ('direction', d), ('action', act), ('log_prob', logp), ('value', val)):
          buf[k][a].append(v)
      env.step()
      for a in range(A):
        buf['reward'][a].append(env.reward[a].float().clone())     # v8 casts to float32 too (v8:311)
      dones.append(env.done.clone())
    out = {k: [torch.stack(v[a]) for a in range(A)] for k, v in buf.items()}
    out['done'] = torch.stack(dones)
    return out

  def _graph_rollout(self, n_steps):
    env, A, N = self.env, self.n_agents, self.env.n_envs
    cache = self.__dict__.setdefault('_graphs', {})
    if n_steps not in cache:
      dev = self.device
      V = env.view_size
      buf = {'image': [torch.zeros((n_steps, N, V, V, 3), dtype=torch.uint8, device=dev) for _ in range(A)],
             'direction': [torch.zeros((n_steps, N), dtype=torch.uint8, device=dev) for _ in range(A)],
             'action': [torch.zeros((n_steps, N), dtype=torch.long, device=dev) for _ in range(A)],
             'log_prob': [torch.zeros((n_steps, N), device=dev) for _ in range(A)],
             'value': [torch.zeros((n_steps, N), device=dev) for _ in range(A)],
             'reward': [torch.zeros((n_steps, N), device=dev) for _ in range(A)],
             'done': torch.zeros((n_steps, N), dtype=torch.uint8, device=dev)}
      t_idx = torch.zeros(1, dtype=torch.long, device=dev)

      def pre():      # policy side of a step: reads the env's obs buffers in place, writes its action buffer
        for a in range(A):
          img, d = env.obs_image[a], env.obs_dir[a]
          act, logp, _, val = self.nets[a].get_action_and_value(img, d)
          env.actions[:N, a] = act.to(torch.uint8)
          for k, v in (('image', img), ('direction', d), ('action', act), ('log_prob', logp), ('value', val)):
            buf[k][a].index_copy_(0, t_idx, v.unsqueeze(0))

      def post():     # after the env kernel: rewards (float32 as at v8:311) and done of the step, then t += 1
        for a in range(A):
          buf['reward'][a].index_copy_(0, t_idx, env.reward[a].float().unsqueeze(0))
        buf['done'].index_copy_(0, t_idx, env.done.unsqueeze(0))
        t_idx.add_(1)

      side = torch.cuda.Stream(device=dev)
      side.wait_stream(torch.cuda.current_stream(dev))
      with torch.cuda.stream(side):     # warm-up outside capture (lazy library initialisation)
        for _ in range(3):
          pre()
          t_idx.zero_()
      torch.cuda.current_stream(dev).wait_stream(side)
      g_pre, g_post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
      with torch.cuda.graph(g_pre):
        pre()
      with torch.cuda.graph(g_post):
        post()
      cache[n_steps] = (g_pre, g_post, buf, t_idx)
    g_pre, g_post, buf, t_idx = cache[n_steps]
    t_idx.zero_()
    for _ in range(n_steps):
      g_pre.replay()
      env.step()          # one launch on the same stream, outside the graphs: its Philox step index is a kernel argument
      g_post.replay()
    return buf

  def update(self, traj, epochs=4, max_grad_norm=0.5):
    """update_agent (v8_robust_ppo.py:295-345) for every agent; returns mean loss per agent."""
    losses = []
    for a in range(self.n_agents):
      adv, ret = compute_gae(traj['reward'][a], traj['value'][a], traj['done'])
      adv, ret = adv.reshape(-1), ret.reshape(-1)
      if adv.std() > 0:                            # v8:316-317
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
      images = traj['image'][a].flatten(0, 1)
      dirs = traj['direction'][a].flatten(0, 1)
      acts = traj['action'][a].flatten(0, 1)
      old_lp = traj['log_prob'][a].flatten(0, 1)
      total = 0.0
      for _ in range(epochs):
        loss = ppo_loss(self.models[a], images, dirs, acts, old_lp, adv, ret)
        self.optimizers[a].zero_grad()
        loss.backward()                            # DDP all-reduces the gradients here
        torch.nn.utils.clip_grad_norm_(self.nets[a].parameters(), max_grad_norm)
        self.optimizers[a].step()
        total += float(loss.detach())
      losses.append(total / epochs)
    return losses
