"""Integer-exact pieces of the input pipeline that sit next to the hot path
(SURVEY.md section 8a rows S1, S2): the rank-sharding sampler and the
nearest-neighbour label resize."""
from .sampler import DistributedSampler, shard_indices      # noqa: F401
from .transforms import nearest_index_table, resize_labels_nearest   # noqa: F401
