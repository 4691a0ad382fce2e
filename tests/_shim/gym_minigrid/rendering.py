"""Rendering is out of scope: the names exist so that the reference's modules import."""


def _unavailable(*a, **k):
  raise NotImplementedError('rendering is not part of the stand-in')


downsample = fill_coords = highlight_img = point_in_circle = point_in_line = point_in_rect = point_in_triangle = rotate_fn = _unavailable
