from timediff_amd.et_diff import EMA, ETDiff  # noqa: F401
