class Error(Exception):
  pass


class RetriesExceededError(Error):
  pass
