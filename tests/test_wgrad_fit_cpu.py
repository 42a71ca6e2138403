"""The per-launch strip-length planner of the halo-staged weight-gradient kernel (hip_backend._fit_tile_strips): a grouped
launch of persistent workgroups should not overflow the chip's 512 slots by a fraction of a round."""
from semseg_amd import hip_backend as hb


class _J:
    def __init__(self, C, H, W):
        self.geom_in = (1, H, W, C)
        self.k, self.stride, self.dil, self.pad, self.cout_pad = (3, 3), 1, 1, 1, C


def _wgs(jobs, fitted, default):
    parts = {48: 1, 64: 1, 96: 3, 192: 12, 384: 48}
    total = 0
    for j in jobs:
        B, H, W, C = j.geom_in
        tiles = B * ((W + 31) // 32) * ((H + 3) // 4)
        s = fitted.get(id(j), default)
        total += -(-tiles // s) * parts[C]
    return total


def test_a_launch_of_ten_48_channel_layers_fits_one_round():
    # ten 48-channel layers at 256 x 256: 512 tiles each; strips of 8 give 640 workgroups = a round and a quarter
    jobs = [_J(48, 256, 256) for _ in range(10)]
    assert _wgs(jobs, {}, 8) == 640
    fitted = hb._fit_tile_strips(jobs, 8)
    assert len(set(fitted.values())) == 1
    s = next(iter(fitted.values()))
    assert 8 < s <= 16 and _wgs(jobs, fitted, 8) <= 512


def test_launches_are_fitted_separately_and_other_jobs_left_alone():
    G = hb._WGRAD_GROUP                                  # layers per grouped launch (csrc/group.h MAXJOBS)
    a = [_J(96, 128, 128) for _ in range(G)]            # one launch of the 96-channel instantiation
    b = [_J(96, 64, 64) for _ in range(4)]              # the next one: 4 x 5 x 3 = 60 at strips of 8
    odd = _J(48, 256, 256)
    odd.stride = 2                                       # not a halo-staged weight gradient: not planned
    fitted = hb._fit_tile_strips(a + b + [odd], 8)
    assert id(odd) not in fitted
    sa, sb = {fitted[id(j)] for j in a}, {fitted[id(j)] for j in b}
    assert len(sa) == 1 and len(sb) == 1
    assert sb == {8}                                     # 60 workgroups fit at the default length
    s = next(iter(sa))
    # no strip length in the planner's range does better (rounds x (strip + fixed cost)), and the launch is not left a
    # fraction of a round over the chip's 512 slots when a longer strip avoids it
    cost = lambda s_: -(-_wgs(a, {id(j): s_ for j in a}, 8) // 512) * (s_ + 2)      # noqa: E731
    assert cost(s) == min(cost(s_) for s_ in range(8, 6 * 8 + 1))
    if G > 16:
        assert s > 8                                     # twice the layers per launch: longer strips, fewer partials
