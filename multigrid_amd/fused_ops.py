"""torch.autograd bindings of include/mgrid_nn.h: the bias / ReLU / pooling passes around the PPO network's two
convolutions (v8_robust_ppo.py:69-75), each as one pass over the activation forward and one backward.

`conv_bias_relu(x, w, b)` and `conv_bias_relu_mean(x, w, b)` are what the trainer calls.  On a GPU with fp32
channels-last activations they run MIOpen's convolution WITHOUT bias followed by the HIP epilogue; anywhere else (CPU
tests, autocast / bf16, other layouts) they are the plain torch expression of the same modules, so results agree to
float rounding of the 25-term mean.  The HIP library is required on a GPU: a missing libmgnn.so raises."""
import ctypes as C
import os

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmgnn.so')

# name -> (restype, argtypes): every symbol include/mgrid_nn.h declares
SYMBOLS = {
    'mgnn_partial_rows': (C.c_int, []),
    'mgnn_bias_relu': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'mgnn_bias_relu_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'mgnn_bias_relu_mean': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    'mgnn_bias_relu_mean_backward': (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
}
_LIB = None


def load():
  global _LIB
  if _LIB is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError('multigrid_amd: %s is missing. Build it with `python -c "import __graft_entry__ as g; g.build()"`.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
      fn = getattr(lib, name)
      fn.restype, fn.argtypes = res, args
    _LIB = lib
  return _LIB


def _check(rc, what):
  if rc != 0:
    raise RuntimeError('%s failed with MGNN error %d' % (what, rc))


def _stream(t):
  return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def fusable(y, bias):
  """The HIP epilogues take fp32 CUDA activations in channels-last storage with a power-of-two channel count."""
  c = y.shape[1] if y.dim() == 4 else 0
  return (y.is_cuda and y.dtype == torch.float32 and bias.dtype == torch.float32 and y.dim() == 4 and c >= 4
          and (c & (c - 1)) == 0 and y.is_contiguous(memory_format=torch.channels_last) and not torch.is_autocast_enabled())


def _partial(y, C_):
  return torch.empty((load().mgnn_partial_rows(), C_), dtype=torch.float32, device=y.device)


class _BiasReLU(torch.autograd.Function):
  """relu(y + bias[None, :, None, None]) in place on the convolution's output."""

  @staticmethod
  def forward(ctx, y, bias):
    N, C_, H, W = y.shape
    with torch.cuda.device(y.device):
      _check(load().mgnn_bias_relu(y.data_ptr(), bias.contiguous().data_ptr(), N * H * W, C_, _stream(y)), 'mgnn_bias_relu')
    ctx.mark_dirty(y)
    ctx.save_for_backward(y)
    return y

  @staticmethod
  def backward(ctx, g):
    out, = ctx.saved_tensors
    N, C_, H, W = out.shape
    g = g.contiguous(memory_format=torch.channels_last)
    grad_in = torch.empty_like(out, memory_format=torch.channels_last)
    part = _partial(out, C_)
    with torch.cuda.device(out.device):
      _check(load().mgnn_bias_relu_backward(out.data_ptr(), g.data_ptr(), grad_in.data_ptr(), part.data_ptr(), N * H * W, C_,
                                            _stream(out)), 'mgnn_bias_relu_backward')
    return grad_in, part.sum(dim=0)


class _BiasReLUMean(torch.autograd.Function):
  """mean over (H, W) of relu(y + bias): [N, C, H, W] -> [N, C]; the activation is read once and never rewritten."""

  @staticmethod
  def forward(ctx, y, bias):
    N, C_, H, W = y.shape
    bias = bias.contiguous()
    out = torch.empty((N, C_), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
      _check(load().mgnn_bias_relu_mean(y.data_ptr(), bias.data_ptr(), out.data_ptr(), N, H * W, C_, _stream(y)), 'mgnn_bias_relu_mean')
    ctx.save_for_backward(y, bias)
    return out

  @staticmethod
  def backward(ctx, g):
    y, bias = ctx.saved_tensors
    N, C_, H, W = y.shape
    g = g.contiguous()
    grad_y = torch.empty_like(y, memory_format=torch.channels_last)
    part = _partial(y, C_)
    with torch.cuda.device(y.device):
      _check(load().mgnn_bias_relu_mean_backward(y.data_ptr(), bias.data_ptr(), g.data_ptr(), grad_y.data_ptr(), part.data_ptr(),
                                                 N, H * W, C_, _stream(y)), 'mgnn_bias_relu_mean_backward')
    return grad_y, part.sum(dim=0)


def bias_relu_(y, bias):
  return _BiasReLU.apply(y, bias)


def bias_relu_mean(y, bias):
  return _BiasReLUMean.apply(y, bias)


def conv_bias_relu(x, weight, bias, padding=1):
  """nn.Conv2d(padding=1) + nn.ReLU (v8_robust_ppo.py:70-71 and :72-73)."""
  if x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
    y = F.conv2d(x, weight, None, padding=padding)
    if fusable(y, bias):
      return bias_relu_(y, bias)
    return F.relu(y + bias.reshape(1, -1, 1, 1))
  return F.relu(F.conv2d(x, weight, bias, padding=padding))


def conv_bias_relu_mean(x, weight, bias, padding=1):
  """nn.Conv2d(padding=1) + nn.ReLU + nn.AdaptiveAvgPool2d((1, 1)) + nn.Flatten (v8_robust_ppo.py:72-75): [N, C]."""
  if x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
    y = F.conv2d(x, weight, None, padding=padding)
    if fusable(y, bias):
      return bias_relu_mean(y, bias)
    return F.relu(y + bias.reshape(1, -1, 1, 1)).mean(dim=(2, 3))
  return F.relu(F.conv2d(x, weight, bias, padding=padding)).mean(dim=(2, 3))
