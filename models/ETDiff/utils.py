from timediff_amd.data import *  # noqa: F401,F403
from timediff_amd.data import TimeSeriesDataset, cycle, default, exists, extract, reverse_normalize  # noqa: F401
from timediff_amd.schedule import cosine_beta_schedule, linear_beta_schedule  # noqa: F401
