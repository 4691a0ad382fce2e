"""gym 0.23.1 seeding: np_random(seed) = a numpy Generator over PCG64(SeedSequence(seed)) whose
deprecated randint(low, high) forwards to integers(low, high)."""
import numpy as np


class RandomNumberGenerator(np.random.Generator):
  def randint(self, low, high=None, size=None, dtype=np.int64):
    return self.integers(low, high, size=size, dtype=dtype)

  def rand(self, *size):
    return self.random(size)


def np_random(seed=None):
  seq = np.random.SeedSequence(seed)
  return RandomNumberGenerator(np.random.PCG64(seq)), seq.entropy
