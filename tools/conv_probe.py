"""Which conv formulations start at once on a fresh box (MIOpen has to find / build a kernel per new shape)."""
import sys, time
import torch
import torch.nn.functional as F
dev = torch.device('cuda', 0)
def stage(name, fn):
  t0 = time.time()
  out = fn()
  torch.cuda.synchronize()
  print('%-40s %.2f s' % (name, time.time() - t0), flush=True)
  return out
mode = sys.argv[1]
B = int(sys.argv[2])
x3 = torch.randn(B, 9, 5, 5, device=dev, requires_grad=True)
w1 = torch.randn(96, 3, 3, 3, device=dev, requires_grad=True); b1 = torch.zeros(96, device=dev)
w2 = torch.randn(192, 32, 3, 3, device=dev, requires_grad=True); b2 = torch.zeros(192, device=dev)
if mode == 'grouped':
  y = stage('grouped conv1 fwd', lambda: F.relu(F.conv2d(x3, w1, b1, padding=1, groups=3)))
  z = stage('grouped conv2 fwd', lambda: F.relu(F.conv2d(y, w2, b2, padding=1, groups=3)))
  stage('grouped bwd', lambda: z.sum().backward())
else:
  ys = []
  for a in range(3):
    y = stage('loop conv1 fwd a%d' % a, lambda: F.relu(F.conv2d(x3[:, 3*a:3*a+3], w1[32*a:32*a+32], b1[32*a:32*a+32], padding=1)))
    ys.append(stage('loop conv2 fwd a%d' % a, lambda: F.relu(F.conv2d(y, w2[64*a:64*a+64], b2[64*a:64*a+64], padding=1))))
  stage('loop bwd', lambda: sum(v.sum() for v in ys).backward())
for rep in range(3):
  t0 = time.time()
  if mode == 'grouped':
    z = F.relu(F.conv2d(F.relu(F.conv2d(x3, w1, b1, padding=1, groups=3)), w2, b2, padding=1, groups=3)); z.sum().backward()
  else:
    tot = 0
    for a in range(3):
      y = F.relu(F.conv2d(x3[:, 3*a:3*a+3], w1[32*a:32*a+32], b1[32*a:32*a+32], padding=1))
      tot = tot + F.relu(F.conv2d(y, w2[64*a:64*a+64], b2[64*a:64*a+64], padding=1)).sum()
    tot.backward()
  torch.cuda.synchronize()
  print('steady fwd+bwd %s B=%d: %.4f s' % (mode, B, time.time() - t0), flush=True)
