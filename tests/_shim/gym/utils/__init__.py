from . import seeding  # noqa: F401
