"""The reference imports this module for its side effects only (envs/gym_multigrid/__init__.py:18)."""
