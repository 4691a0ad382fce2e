import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def cluttered_cells():
  from multigrid_amd import layouts
  return layouts.cluttered_fixed_15x15()


@pytest.fixture(scope='session')
def empty_cells():
  """Appendix-B synthetic map: border walls, goal (13,13), no clutter."""
  from multigrid_amd import layouts
  return layouts.empty_bordered(15, 15, goal=(13, 13))


@pytest.fixture(scope='session')
def playground_cells():
  """Synthetic 9x9 map with keys, balls, boxes, doors, floor, lava and goals (mutable-grid tests)."""
  from multigrid_amd import layouts
  return layouts.object_playground()
