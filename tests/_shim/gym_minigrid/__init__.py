"""Stand-in for the slice of gym-minigrid 1.0.3 the reference touches (tests/_shim/README.md).  Not gym-minigrid."""
