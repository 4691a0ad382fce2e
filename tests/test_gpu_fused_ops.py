"""The PPO trainer's fused convolution epilogues (include/mgrid_nn.h, multigrid_amd/csrc/nn_ops.hip) against the torch
modules they replace (v8_robust_ppo.py:69-75): forward values, input gradients and bias gradients, fp32."""
import pytest
import torch
import torch.nn.functional as F

from multigrid_amd import fused_ops, ppo

pytestmark = pytest.mark.gpu


def _act(n, c, hw, seed):
  g = torch.Generator().manual_seed(seed)
  y = torch.randn(n, c, hw, hw, generator=g).cuda().contiguous(memory_format=torch.channels_last)
  b = (0.3 * torch.randn(c, generator=g)).cuda()
  return y, b


@pytest.mark.parametrize('n,c,hw', [(1, 32, 5), (1000, 32, 5), (4097, 64, 5), (513, 64, 7), (64, 4, 3), (0, 32, 5)])
def test_bias_relu_forward_and_backward(n, c, hw):
  y, b = _act(n, c, hw, 1)
  yr, br = y.clone().requires_grad_(True), b.clone().requires_grad_(True)
  ref = F.relu(yr + br.reshape(1, -1, 1, 1))
  yf, bf = y.clone().requires_grad_(True), b.clone().requires_grad_(True)
  assert fused_ops.fusable(yf, bf) or n == 0
  out = fused_ops.bias_relu_(yf * 1.0, bf)                 # (a non-leaf, like a convolution's output: modified in place)
  assert torch.equal(out, ref)                             # add then max: the same two float operations
  if n == 0:
    return
  w = torch.randn_like(ref)
  (ref * w).sum().backward()
  (out * w).sum().backward()
  assert torch.equal(yf.grad, yr.grad)
  assert torch.allclose(bf.grad, br.grad, rtol=1e-5, atol=1e-4 * max(1.0, n ** 0.5))   # a sum of n*hw*hw terms, other order


@pytest.mark.parametrize('n,c,hw', [(1, 64, 5), (1000, 64, 5), (4097, 32, 5), (513, 64, 7), (2, 8, 3)])
def test_bias_relu_mean_forward_and_backward(n, c, hw):
  y, b = _act(n, c, hw, 2)
  yr, br = y.clone().requires_grad_(True), b.clone().requires_grad_(True)
  ref = F.relu(yr + br.reshape(1, -1, 1, 1)).mean(dim=(2, 3))
  yf, bf = y.clone().requires_grad_(True), b.clone().requires_grad_(True)
  out = fused_ops.bias_relu_mean(yf, bf)
  assert out.shape == (n, c) and torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
  w = torch.randn_like(ref)
  (ref * w).sum().backward()
  (out * w).sum().backward()
  assert torch.allclose(yf.grad, yr.grad, rtol=1e-6, atol=1e-7)
  assert yf.grad.is_contiguous(memory_format=torch.channels_last)
  assert torch.allclose(bf.grad, br.grad, rtol=1e-5, atol=1e-5 * max(1.0, n ** 0.5))
  assert torch.equal(y, yf.detach())                       # the activation is only read


def test_bias_gradient_is_the_same_every_run():
  y, b = _act(3000, 64, 5, 3)
  grads = []
  for _ in range(3):
    yf, bf = y.clone().requires_grad_(True), b.clone().requires_grad_(True)
    fused_ops.bias_relu_mean(yf, bf).square().sum().backward()
    grads.append(bf.grad.clone())
  assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


def test_other_layouts_and_dtypes_take_the_torch_expression():
  y, b = _act(16, 32, 5, 4)
  assert not fused_ops.fusable(y.contiguous(), b)                       # NCHW storage
  assert not fused_ops.fusable(y.to(torch.bfloat16), b.to(torch.bfloat16))
  assert not fused_ops.fusable(_act(16, 96, 5, 4)[0], _act(16, 96, 5, 4)[1])   # the grouped convolution's 3 x 32 channels
  x = torch.randn(16, 3, 5, 5).cuda()
  w = torch.randn(32, 3, 3, 3).cuda()
  assert torch.allclose(fused_ops.conv_bias_relu(x, w, b), F.relu(F.conv2d(x, w, b, padding=1)), atol=1e-5)


def test_stacked_nets_on_the_gpu_equal_the_per_agent_reference_nets():
  """StackedAgentNets (fused epilogues) against AgentNet (nn.Sequential of the reference's modules) on the same weights:
  logits, values and every parameter gradient, agent per agent."""
  torch.manual_seed(0)
  st = ppo.StackedAgentNets(3).cuda()
  img = torch.randint(0, 11, (3, 257, 5, 5, 3), dtype=torch.uint8).cuda()
  d = torch.randint(0, 4, (3, 257)).cuda()
  logits, values = st(img, d)
  (logits.square().sum() + values.sum()).backward()
  for a in range(3):
    net = st.agent_net(a)
    l, v = net(img[a], d[a])
    assert torch.allclose(logits[a], l, atol=2e-6) and torch.allclose(values[a], v.reshape(-1), atol=2e-6)
    (l.square().sum() + v.sum()).backward()
    sd = dict(net.named_parameters())
    for ref_name, name, _ in ppo._STACK:
      got, want = getattr(st, name).grad[a], sd[ref_name].grad.reshape(getattr(st, name).grad[a].shape)
      assert torch.allclose(got, want, rtol=1e-4, atol=2e-5), (a, ref_name, (got - want).abs().max().item())
