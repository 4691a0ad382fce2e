"""K1-K14 (tests/kat_cases.py) against BOTH CPU oracles."""
import pytest

from oracle_adapters import KINDS
from kat_cases import *  # noqa: F401,F403  (the test functions)


@pytest.fixture(params=['c', 'py'])
def make(request):
  return KINDS[request.param]
