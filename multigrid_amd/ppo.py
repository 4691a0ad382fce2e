"""Batched v8 PPO on top of the device-resident env.

PyTorch-ROCm only (no HIP kernels here): the policy consumes `env.obs_image` / `env.obs_dir` in place, writes its
sampled actions into `env.actions`, and `env.step()` is one kernel launch on the same stream — no host round trip per
step, where the reference does 3 batch-1 forwards with an `.item()` sync each (v8_robust_ppo.py:138-157).

The A per-agent networks stay independent (own weights, own Adam state, own gradient clipping, own advantage
normalisation — v8_robust_ppo.py:117-121, 295-345) but are EVALUATED TOGETHER: `StackedAgentNets` keeps every
parameter with a leading agent axis and runs one grouped convolution / batched matmul for all agents, so a rollout step
and an update epoch are one forward (and one backward) instead of A.

Reference anchors (relative to /root/reference):
  v8_robust_ppo.py:35-102   SimplePPOAgent (architecture, init, get_action_and_value)
  v8_robust_ppo.py:275-293  compute_gae   (gamma .99, lambda .95; next_value = 0 at the last step)
  v8_robust_ppo.py:295-345  update_agent  (4 full-batch epochs, clip [0.8, 1.2], vf .5, ent .02,
                                           grad-clip .5, advantages normalised when std > 0)
  v8_robust_ppo.py:117-121  one independent net + Adam(lr=1e-4) per agent
  v8_robust_ppo.py:233-271  run_episode's bookkeeping (EpisodeStats)
  networks/multigrid_network.py:8-79  MultiGridNetwork (BASELINE config 5's policy; policy='multigrid_net')
Module / parameter names match the reference's, so a reference `state_dict` checkpoint (v8_robust_ppo.py:362-366) loads
with `load_state_dict(torch.load(path, weights_only=True))`.
"""
import torch
from torch import nn
from torch.distributions import Categorical
import torch.nn.functional as F

from . import fused_ops


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

  @staticmethod
  def _valid_conv(x, conv):
    """nn.Conv2d without padding as im2col + ONE matmul (same arithmetic, same parameters).  Neither MIOpen (the 3x3 /
    1x1 output maps of this network are shapes it may have to compile for minutes on a fresh box) nor F.unfold (ATen's
    im2col launches one kernel per sample on ROCm: 17 ms for 4096 samples) is involved: the kh*kw shifted windows are
    sliced and stacked — a handful of batched copies."""
    B, C, H, W = x.shape
    kh, kw = conv.kernel_size
    Ho, Wo = H - kh + 1, W - kw + 1
    win = torch.stack([x[:, :, ky:ky + Ho, kx:kx + Wo] for ky in range(kh) for kx in range(kw)], dim=2)   # [B, C, kh*kw, Ho, Wo]
    cols = win.reshape(B, C * kh * kw, Ho * Wo).transpose(1, 2)            # (c, ky, kx) order = the weight's
    out = F.linear(cols, conv.weight.reshape(conv.out_channels, -1), conv.bias)                            # [B, Ho*Wo, O]
    return out.transpose(1, 2).reshape(B, conv.out_channels, Ho, Wo)

  def features(self, image, directions):
    """The 64 features under the last layer (multigrid_network.py:75-78 without head[4])."""
    x = image.permute(0, 3, 1, 2).float()
    for layer in self.image_layers:
      x = self._valid_conv(x, layer) if isinstance(layer, nn.Conv2d) else layer(x)
    onehot = F.one_hot(directions.long(), num_classes=4).reshape(image.shape[0], -1).float()
    return self.head[:-1](torch.cat([x, self.direction_layers(onehot)], dim=-1))

  def forward(self, image, directions):
    """image [B, V, V, 3]; directions [B, n_agents] (every agent's compass, multigrid_network.py:70-74)."""
    return self.head[-1](self.features(image, directions))


class MultiGridPPONet(MultiGridNet):
  """MultiGridNetwork as a PPO policy (BASELINE config 5 names it as the policy; the reference defines it as a
  Q-network and no trainer in its tree uses it): its n_actions outputs are the actor's logits, and a `critic`
  Linear(64, 1) on the same 64 penultimate features gives the value PPO needs.  Every reference parameter keeps its
  name, so a MultiGridNetwork state_dict loads with strict=False (only `critic.*` is new)."""

  def __init__(self, n_actions=7, n_agents=3, **kw):
    super().__init__(n_actions=n_actions, n_agents=n_agents, **kw)
    self.critic = nn.Linear(64, 1)

  def forward(self, image, directions):
    h = self.features(image, directions)
    return self.head[-1](h), self.critic(h).squeeze(-1)


# reference parameter name -> (stacked parameter, shape without the agent axis)
_STACK = (('image_conv.0.weight', 'conv1_w', (32, 3, 3, 3)), ('image_conv.0.bias', 'conv1_b', (32,)),
          ('image_conv.2.weight', 'conv2_w', (64, 32, 3, 3)), ('image_conv.2.bias', 'conv2_b', (64,)),
          ('direction_embed.weight', 'embed', (4, 8)),
          ('shared.0.weight', 'fc1_w', (128, 72)), ('shared.0.bias', 'fc1_b', (128,)),
          ('shared.2.weight', 'fc2_w', (64, 128)), ('shared.2.bias', 'fc2_b', (64,)),
          ('actor.weight', 'actor_w', (7, 64)), ('actor.bias', 'actor_b', (7,)),
          ('critic.weight', 'critic_w', (1, 64)), ('critic.bias', 'critic_b', (1,)))


class StackedAgentNets(nn.Module):
  """A independent SimplePPOAgents (v8_robust_ppo.py:35-102, 117-121) evaluated together.

  Every parameter carries a leading agent axis; the linear layers run as batched matmuls over the agent axis, the
  convolutions per agent (conv_mode='loop', the default: exactly the two conv shapes of the reference net, which MIOpen
  serves at once) or as ONE grouped convolution (conv_mode='grouped', groups = A: agent a's channels only ever meet agent
  a's filters — fewer launches, but a shape MIOpen compiles for minutes on a fresh box before its first call returns).
  Mathematically both are exactly A separate nets: gradients of a summed loss do not mix agents."""

  def __init__(self, n_agents, n_actions=7, conv_mode='loop'):
    super().__init__()
    self.n_agents, self.n_actions, self.conv_mode = n_agents, n_actions, conv_mode
    nets = [AgentNet(n_actions) for _ in range(n_agents)]        # the reference's initialisation, agent by agent
    for ref_name, name, _ in _STACK:
      self.register_parameter(name, nn.Parameter(torch.stack([n.state_dict()[ref_name] for n in nets]).contiguous()))

  def agent_state_dict(self, a):
    """Agent a's weights under the reference's names (what SimplePPOAgent.state_dict() holds)."""
    return {ref_name: getattr(self, name)[a].detach().clone() for ref_name, name, _ in _STACK}

  def load_agent_state_dict(self, a, sd):
    with torch.no_grad():
      for ref_name, name, shape in _STACK:
        getattr(self, name)[a].copy_(sd[ref_name].reshape(shape))

  def agent_net(self, a):
    """A stand-alone AgentNet holding a COPY of agent a's weights."""
    net = AgentNet(self.n_actions)
    net.load_state_dict(self.agent_state_dict(a))
    return net.to(self.conv1_w.device)

  def agent_parameters(self, a):
    return [getattr(self, name)[a] for _, name, _ in _STACK]

  def forward(self, image, direction):
    """image [A, B, V, V, 3] (uint8 or float, [x][y][c] as the env writes it), direction [A, B] ->
    (logits [A, B, n_actions], values [A, B])."""
    A, B = image.shape[0], image.shape[1]
    V = image.shape[2]
    if self.conv_mode == 'grouped':
      x = image.permute(1, 0, 4, 2, 3).reshape(B, A * 3, V, V).float()      # v8:75 per agent, agents side by side
      x = F.relu(F.conv2d(x, self.conv1_w.reshape(A * 32, 3, 3, 3), self.conv1_b.reshape(-1), padding=1, groups=A))
      x = F.relu(F.conv2d(x, self.conv2_w.reshape(A * 64, 32, 3, 3), self.conv2_b.reshape(-1), padding=1, groups=A))
      feat = x.mean(dim=(2, 3)).reshape(B, A, 64).transpose(0, 1)           # AdaptiveAvgPool2d((1,1)) + Flatten
    else:                                                                   # 'loop': one ordinary convolution per agent
      outs = []
      for a in range(A):
        # bias + ReLU (+ the global average) are one pass over the activation each way (fused_ops.py, csrc/nn_ops.hip)
        y = fused_ops.conv_bias_relu(image[a].permute(0, 3, 1, 2).float(), self.conv1_w[a], self.conv1_b[a])   # v8:75
        outs.append(fused_ops.conv_bias_relu_mean(y, self.conv2_w[a], self.conv2_b[a]))
      feat = torch.stack(outs)
    idx = direction.long().unsqueeze(-1).expand(A, B, 8)
    emb = torch.gather(self.embed, 1, idx)                                  # direction_embed, per agent
    h = torch.cat([feat, emb], dim=-1)                                      # [A, B, 72]
    h = F.relu(torch.baddbmm(self.fc1_b.unsqueeze(1), h, self.fc1_w.transpose(1, 2)))
    h = F.relu(torch.baddbmm(self.fc2_b.unsqueeze(1), h, self.fc2_w.transpose(1, 2)))
    logits = torch.baddbmm(self.actor_b.unsqueeze(1), h, self.actor_w.transpose(1, 2))
    values = torch.baddbmm(self.critic_b.unsqueeze(1), h, self.critic_w.transpose(1, 2)).squeeze(-1)
    return logits, values

  def clip_grad_norm_(self, max_norm):
    """torch.nn.utils.clip_grad_norm_(agent.parameters(), max_norm) (v8:340) for every agent separately."""
    grads = [getattr(self, name).grad for _, name, _ in _STACK]
    sq = sum(g.reshape(self.n_agents, -1).pow(2).sum(dim=1) for g in grads if g is not None)
    coef = (max_norm / (sq.sqrt() + 1e-6)).clamp(max=1.0)
    for g in grads:
      if g is not None:
        g.mul_(coef.reshape((-1,) + (1,) * (g.dim() - 1)))


class PerAgentNets(nn.Module):
  """A separate networks behind the stacked interface (images [A, B, ...], directions [A, B]) -> ([A, B, n_actions],
  [A, B]).  all_dirs: each net also sees the other agents' directions (MultiGridNetwork, multigrid_network.py:70-74)."""

  def __init__(self, nets, all_dirs=False):
    super().__init__()
    self.nets = nn.ModuleList(nets)
    self.n_agents, self.all_dirs = len(nets), all_dirs

  def forward(self, image, direction):
    d_all = direction.transpose(0, 1) if self.all_dirs else None
    outs = [net(image[a], d_all if self.all_dirs else direction[a]) for a, net in enumerate(self.nets)]
    return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])

  def agent_state_dict(self, a):
    return {k: v.detach().clone() for k, v in self.nets[a].state_dict().items()}

  def load_agent_state_dict(self, a, sd):
    self.nets[a].load_state_dict(sd, strict=False)      # a reference MultiGridNetwork checkpoint has no critic.*

  def agent_net(self, a):
    return self.nets[a]

  def agent_parameters(self, a):
    return list(self.nets[a].parameters())

  def clip_grad_norm_(self, max_norm):
    for net in self.nets:
      torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)


def compute_gae(rewards, values, dones, gamma=0.99, gae_lambda=0.95):
  """compute_gae (v8_robust_ppo.py:275-293) for every env (and agent) at once, as a parallel scan over time.

  rewards, values: [T, ...]; dones: bool/0-1 [T] or broadcastable to rewards after the time axis ([T, N] against
  [T, A, N] is accepted).  Returns (advantages, returns) in float64 — the reference runs this recurrence on Python
  floats and casts to float32 afterwards (v8:308-310).  Quirks kept: the value after the last stored step is 0 (no
  bootstrap), and `done` of step t gates the bootstrap from step t+1.

  adv[t] = delta[t] + c[t] * adv[t+1] with c[t] = gamma * lambda * (1 - done[t]) is a first-order linear recurrence;
  pairs (c, d) compose associatively — (c1, d1) then (c2, d2) = (c1 * c2, d1 + c1 * d2) — so ceil(log2 T) doubling
  steps of whole-tensor arithmetic replace the T-step Python loop (and its ~6 T kernel launches)."""
  T = rewards.shape[0]
  r, v = rewards.double(), values.double()
  d = dones.to(torch.float64)
  if d.dim() == 1:
    d = d.reshape((T,) + (1,) * (r.dim() - 1))
  elif d.dim() < r.dim():
    d = d.unsqueeze(1)                              # [T, N] against [T, A, N]
  d = d.expand_as(r)
  nonterm = 1.0 - d
  next_v = torch.cat([v[1:], torch.zeros_like(v[:1])], dim=0)
  acc = r + gamma * next_v * nonterm - v            # delta (v8:285)
  coef = gamma * gae_lambda * nonterm
  step = 1
  while step < T:                                   # acc[t] covers steps t .. t + step - 1, coef[t] their product
    acc = torch.cat([acc[:-step] + coef[:-step] * acc[step:], acc[-step:]], dim=0)
    coef = torch.cat([coef[:-step] * coef[step:], coef[-step:]], dim=0)
    step *= 2
  return acc, acc + v


def ppo_loss_terms(logits, new_values, actions, old_log_probs, advantages, returns,
                   clip_low=0.8, clip_high=1.2, vf_coef=0.5, ent_coef=0.02):
  """The loss of one epoch of update_agent (v8_robust_ppo.py:326-336) from the policy's outputs; means are taken over
  the LAST axis, so [B] inputs give a scalar and stacked [A, B] inputs one loss per agent."""
  logp_all = F.log_softmax(logits, dim=-1)
  new_log_probs = logp_all.gather(-1, actions.long().unsqueeze(-1)).squeeze(-1)
  entropy = -(logp_all.exp() * logp_all).sum(dim=-1)
  ratio = torch.exp(new_log_probs - old_log_probs)
  surr1 = ratio * advantages
  surr2 = torch.clamp(ratio, clip_low, clip_high) * advantages
  policy_loss = -torch.min(surr1, surr2).mean(dim=-1)
  value_loss = vf_coef * (new_values - returns).pow(2).mean(dim=-1)
  entropy_loss = -ent_coef * entropy.mean(dim=-1)
  return policy_loss + value_loss + entropy_loss


def ppo_loss(model, images, directions, actions, old_log_probs, advantages, returns, **kw):
  """`model(images, directions)` -> (logits, values): an AgentNet or its DistributedDataParallel wrapper."""
  logits, new_values = model(images, directions)
  return ppo_loss_terms(logits, new_values, actions, old_log_probs, advantages, returns, **kw)


class EpisodeStats:
  """run_episode's bookkeeping (v8_robust_ppo.py:233-271) for N envs at once, on device: per-agent return of the
  episode in progress, goal touches (shaped reward >= 4.5, v8:246-248), and sums over finished episodes.
  `add` takes a whole [T, A, N] rollout in one go: cumulative sums along time and one gather at every env's last
  `done` — no Python loop over steps, no boolean-mask indexing, no host synchronisation.
  `reduce()` adds the finished-episode sums over all ranks (one small all-reduce: RCCL on GPUs, gloo in tests)."""

  def __init__(self, n_agents, n_envs, device):
    self.ep_return = torch.zeros((n_agents, n_envs), dtype=torch.float64, device=device)
    self.ep_touches = torch.zeros(n_envs, dtype=torch.int64, device=device)
    # finished episodes: [count, sum of collective returns, sum of goal touches]
    self.totals = torch.zeros(3, dtype=torch.float64, device=device)

  @torch.no_grad()
  def add(self, reward, done):
    """reward [A, N] (or [T, A, N]), done [N] (or [T, N]) of one step (or T steps in order)."""
    if reward.dim() == 2:
      reward, done = reward.unsqueeze(0), done.unsqueeze(0)
    T = reward.shape[0]
    r = reward.to(torch.float64)
    fin = done.to(torch.bool)                                        # [T, N]
    cum = torch.cumsum(r, dim=0)                                     # [T, A, N] return since the rollout began
    tcum = torch.cumsum((reward >= 4.5).sum(dim=1), dim=0)           # [T, N] goal touches since the rollout began
    # every env's last finished step (or none): episodes tile [0, last], so their returns add up to the carried
    # return plus everything up to `last`; what follows `last` is the new episode in progress
    steps = torch.arange(1, T + 1, device=fin.device).unsqueeze(1)
    last = (fin * steps).amax(dim=0)                                 # 0 = no episode finished, else 1 + step index
    any_fin = last > 0
    li = (last - 1).clamp(min=0)
    upto = torch.gather(cum, 0, li.reshape(1, 1, -1).expand(1, cum.shape[1], -1))[0]       # [A, N]
    tupto = torch.gather(tcum, 0, li.reshape(1, -1))[0]                                    # [N]
    zero = torch.zeros((), dtype=torch.float64, device=fin.device)
    self.totals[0] += fin.sum()
    self.totals[1] += torch.where(any_fin, (self.ep_return + upto).sum(dim=0), zero).sum()   # collective = sum over agents (v8:425)
    self.totals[2] += torch.where(any_fin, self.ep_touches + tupto, torch.zeros_like(tupto)).sum()
    self.ep_return = torch.where(any_fin, cum[-1] - upto, self.ep_return + cum[-1])
    self.ep_touches = torch.where(any_fin, tcum[-1] - tupto, self.ep_touches + tcum[-1])

  def reduce(self, across_ranks=True):
    """{'episodes', 'mean_collective_return', 'goal_touches'} over every rank's finished episodes."""
    import torch.distributed as dist
    t = self.totals.clone()
    if across_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
      all_reduce_(t)
    n = float(t[0])
    return {'episodes': int(n), 'mean_collective_return': float(t[1]) / n if n else 0.0, 'goal_touches': int(t[2])}


def _through_host(t):
  """gloo takes device tensors, but on this image its device path stands still for anything beyond a few bytes (round 3:
  two ranks on one GPU, a 450 KB gradient buffer, both ranks inside all_reduce; the three-double statistics went
  through).  The rehearsal backend therefore moves device tensors through host memory; RCCL ('nccl') never does."""
  import torch.distributed as dist
  return t.is_cuda and dist.get_backend() != 'nccl'


def all_reduce_(t):
  """dist.all_reduce (sum) in place, on whatever backend the process group runs."""
  import torch.distributed as dist
  if _through_host(t):
    h = t.cpu()
    dist.all_reduce(h)
    t.copy_(h)
  else:
    dist.all_reduce(t)
  return t


def broadcast_(t, src=0):
  import torch.distributed as dist
  if _through_host(t):
    h = t.cpu()
    dist.broadcast(h, src)
    t.copy_(h)
  else:
    dist.broadcast(t, src)
  return t


class BatchedController:
  """SimpleController (v8_robust_ppo.py:105-366) over N envs at once.

  `env` is a BatchedMultiGrid (or anything with its tensors: obs_image [A,N,V,V,3], obs_dir [A,N], reward [A,N],
  done [N], actions [stride,4], n_envs, n_agents, step(), reset()).
  policy='simple': the v8 script's SimplePPOAgent per agent (StackedAgentNets); policy='multigrid_net': the
  reference's MultiGridNetwork per agent (networks/multigrid_network.py, BASELINE config 5), fed every agent's
  direction, with a value head added for PPO (MultiGridPPONet).
  With torch.distributed initialised (ddp, default: on when there are several ranks) the gradients of all A nets are
  averaged across ranks in ONE flat-buffer all-reduce per epoch, issued by update() right after backward (RCCL over
  xGMI on GPUs, gloo in the rehearsals and CPU tests); envs stay sharded and nothing else crosses ranks.  Not
  DistributedDataParallel: the whole gradient is ~450 KB, so bucketing and overlap with backward have nothing to hide,
  and its reducer, driving gloo on device buckets from autograd's hooks, is where the two-rank rehearsal on one GPU
  stood still in round 2 (both ranks inside loss.backward()).  amp=torch.bfloat16 runs the update's forward / backward under autocast — NOT the
  reference's precision (fp32, the default here); it exists to show what the matrix cores would buy.
  """

  def __init__(self, env, lr=1e-4, device=None, ddp=None, seed=0, policy='simple', conv_mode='loop', amp=None):
    self.env = env
    self.n_agents = A = env.n_agents
    self.device = device if device is not None else env.obs_image.device
    self.amp = amp
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    torch.manual_seed(seed)                       # same initial weights on every rank
    if policy == 'simple':
      self.policy = StackedAgentNets(A, conv_mode=conv_mode).to(self.device)
    elif policy == 'multigrid_net':
      self.policy = PerAgentNets([MultiGridPPONet(n_agents=A) for _ in range(A)], all_dirs=True).to(self.device)
    else:
      raise ValueError('policy must be "simple" or "multigrid_net"')
    self.policy_kind = policy
    # action sampling must differ between ranks (same weights, different envs): re-seed the sampler per rank
    torch.manual_seed(seed + 1 + (dist.get_rank() if distributed else 0))
    if ddp is None:
      ddp = distributed
    self.ddp = bool(ddp)
    self.model = self.policy
    if self.ddp:       # every rank starts from rank 0's weights (same seed already gives that; this makes it hold by construction)
      flat = torch._utils._flatten_dense_tensors([p.data for p in self.policy.parameters()])
      broadcast_(flat, 0)
      for p, f in zip(self.policy.parameters(), torch._utils._unflatten_dense_tensors(flat, [p.data for p in self.policy.parameters()])):
        p.data.copy_(f)
    # one Adam over every agent's parameters = one Adam per agent (the update is elementwise), v8:121
    self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=lr)
    self.stats = EpisodeStats(A, env.n_envs, self.device)

  @property
  def nets(self):
    """Per-agent networks: for policy='simple' stand-alone AgentNet COPIES of the stacked weights."""
    return [self.policy.agent_net(a) for a in range(self.n_agents)]

  def save_models(self, path_prefix):
    """save_models (v8_robust_ppo.py:362-366): one state_dict per agent, the reference's file names and keys."""
    import os
    os.makedirs(os.path.dirname(path_prefix) or '.', exist_ok=True)
    for i in range(self.n_agents):
      torch.save(self.policy.agent_state_dict(i), '%s_agent_%d.pth' % (path_prefix, i))

  def load_models(self, path_prefix):
    """Loads files written by save_models — or by the reference: the parameter names are the same."""
    for i in range(self.n_agents):
      self.policy.load_agent_state_dict(i, torch.load('%s_agent_%d.pth' % (path_prefix, i), map_location=self.device,
                                                      weights_only=True))

  def _act(self, image, direction):
    logits, values = self.policy(image, direction)
    dist = Categorical(logits=logits, validate_args=False)   # argument validation synchronises with the host
    action = dist.sample()
    return action, dist.log_prob(action), values

  @torch.no_grad()
  def rollout(self, n_steps, use_graph=None):
    """run_episode (v8_robust_ppo.py:222-273) for N envs x n_steps, everything on device.
    Returns {'image' [T, A, N, V, V, 3], 'direction' [T, A, N], 'action', 'log_prob', 'value', 'reward' [T, A, N]
    (float64, like the Python floats v8's compute_gae runs on, v8:275-293; advantages() casts afterwards as v8:309-311 do), 'done' [T, N]}.

    use_graph (default: on for CUDA devices): the per-step policy work — one stacked forward, sampling, the action
    write and the trajectory stores — is launch-bound at these sizes, so it is captured once into two HIP graphs and
    replayed around the env's own kernel launch: a step is then graph replay + `mg_step` + graph replay.  The returned
    tensors are the graph's own buffers: they are overwritten by the next graphed rollout of the same length."""
    if use_graph is None:
      use_graph = self.device.type == 'cuda'
    if use_graph:
      return self._graph_rollout(n_steps)
    env, A, N = self.env, self.n_agents, self.env.n_envs
    buf = {k: [] for k in ('image', 'direction', 'action', 'log_prob', 'value', 'reward', 'done')}
    for _ in range(n_steps):
      img, d = env.obs_image.clone(), env.obs_dir.clone()
      act, logp, val = self._act(img, d)
      env.actions[:N, :A] = act.t().to(torch.uint8)
      env.step()
      for k, v in (('image', img), ('direction', d), ('action', act), ('log_prob', logp), ('value', val),
                   ('reward', env.reward.double().clone()), ('done', env.done.clone())):
        buf[k].append(v)
    return {k: torch.stack(v) for k, v in buf.items()}

  def _graph_rollout(self, n_steps):
    env, A, N = self.env, self.n_agents, self.env.n_envs
    cache = self.__dict__.setdefault('_graphs', {})
    if n_steps not in cache:
      dev = self.device
      V = env.view_size
      buf = {'image': torch.zeros((n_steps, A, N, V, V, 3), dtype=torch.uint8, device=dev),
             'direction': torch.zeros((n_steps, A, N), dtype=torch.uint8, device=dev),
             'action': torch.zeros((n_steps, A, N), dtype=torch.long, device=dev),
             'log_prob': torch.zeros((n_steps, A, N), device=dev),
             'value': torch.zeros((n_steps, A, N), device=dev),
             'reward': torch.zeros((n_steps, A, N), dtype=torch.float64, device=dev),
             'done': torch.zeros((n_steps, N), dtype=torch.uint8, device=dev)}
      t_idx = torch.zeros(1, dtype=torch.long, device=dev)

      def pre():      # policy side of a step: reads the env's obs buffers in place, writes its action buffer
        img, d = env.obs_image, env.obs_dir
        act, logp, val = self._act(img, d)
        env.actions[:N, :A] = act.t().to(torch.uint8)
        for k, v in (('image', img), ('direction', d), ('action', act), ('log_prob', logp), ('value', val)):
          buf[k].index_copy_(0, t_idx, v.unsqueeze(0))

      def post():     # after the env kernel: rewards (float64: GAE runs on them uncast, v8:275-293) and done of the step, then t += 1
        buf['reward'].index_copy_(0, t_idx, env.reward.double().unsqueeze(0))
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

  def _allreduce_gradients(self):
    """Mean of every parameter's gradient over the ranks: one flat buffer, one all-reduce."""
    import torch.distributed as dist
    grads = [p.grad for p in self.policy.parameters() if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    all_reduce_(flat)
    flat.div_(dist.get_world_size())
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
      g.copy_(f)

  def advantages(self, traj):
    """GAE per agent and env (v8:275-293), then the per-agent normalisation of update_agent (v8:308-317): float32
    tensors [A, T*N] (advantages, returns)."""
    A = self.n_agents
    adv, ret = compute_gae(traj['reward'], traj['value'], traj['done'])        # [T, A, N] float64
    adv = adv.float().transpose(0, 1).reshape(A, -1)                           # torch.FloatTensor(advantages), v8:309
    ret = ret.float().transpose(0, 1).reshape(A, -1)
    std = adv.std(dim=1, keepdim=True)
    adv = torch.where(std > 0, (adv - adv.mean(dim=1, keepdim=True)) / (std + 1e-8), adv)   # v8:316-317
    return adv, ret

  def update(self, traj, epochs=4, max_grad_norm=0.5, as_tensor=False):
    """update_agent (v8_robust_ppo.py:295-345) for every agent at once: `epochs` full-batch passes; one forward and
    one backward per pass for all agents (their losses are summed — no parameter is shared, so every agent receives
    exactly its own gradient), per-agent gradient clipping, one optimizer step.  Returns the mean loss per agent
    (a list of floats: ONE host synchronisation at the end; as_tensor=True returns the device tensor instead)."""
    A = self.n_agents
    adv, ret = self.advantages(traj)
    images = traj['image'].transpose(0, 1).flatten(1, 2)                       # [A, T*N, V, V, 3]
    dirs = traj['direction'].transpose(0, 1).flatten(1, 2)
    acts = traj['action'].transpose(0, 1).flatten(1, 2)
    old_lp = traj['log_prob'].transpose(0, 1).flatten(1, 2)
    losses = []
    for _ in range(epochs):
      if self.amp is not None:
        with torch.autocast(device_type=self.device.type, dtype=self.amp):
          logits, values = self.model(images, dirs)
        logits, values = logits.float(), values.float()
      else:
        logits, values = self.model(images, dirs)
      per_agent = ppo_loss_terms(logits, values, acts, old_lp, adv, ret)       # [A]
      self.optimizer.zero_grad(set_to_none=True)
      per_agent.sum().backward()
      if self.ddp:
        self._allreduce_gradients()
      self.policy.clip_grad_norm_(max_grad_norm)
      self.optimizer.step()
      losses.append(per_agent.detach())
    mean = torch.stack(losses).mean(dim=0)
    return mean if as_tensor else [float(x) for x in mean.cpu()]
