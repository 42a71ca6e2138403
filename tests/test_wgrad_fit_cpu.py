"""The per-launch strip-length planner of the halo-staged weight-gradient kernels (hip_backend._fit_tile_strips): a grouped
launch of persistent workgroups should not overflow the workgroups the chip holds at once by a fraction of a round.  The
geometry (workgroups per strip, resident workgroups, which instantiation) comes from the library
(ssa_conv2d_wgrad_tile_geometry, csrc/conv_wgrad_tile.hip): one workgroup per CU -- the 4-wave kernels hold more than 256
registers per lane --, two halves of the n-blocks per (96 co x 96 ci) block on the all-taps form."""
import ctypes

from semseg_amd import hip_backend as hb


class _J:
    def __init__(self, C, H, W):
        self.geom_in = (1, H, W, C)
        self.k, self.stride, self.dil, self.pad, self.cout_pad = (3, 3), 1, 1, 1, C


def _geom(C):
    parts, slots, kind = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    assert hb.lib().ssa_conv2d_wgrad_tile_geometry(C, C, ctypes.byref(parts), ctypes.byref(slots), ctypes.byref(kind)) == 0
    return parts.value, slots.value, kind.value


def _wgs(jobs, fitted, default):
    total = 0
    for j in jobs:
        B, H, W, C = j.geom_in
        tiles = B * ((W + 31) // 32) * ((H + 3) // 4)
        s = fitted.get(id(j), default)
        total += -(-tiles // s) * _geom(C)[0]
    return total


def test_geometry_of_the_instantiations():
    assert _geom(48) == (1, 256, 48) and _geom(64) == (1, 256, 64)
    # the all-taps form: 2 halves x (C / 96)^2 blocks; one kind -> one grouped launch for 96 / 192 / 384 channels
    assert _geom(96) == (2, 256, 0) and _geom(192) == (8, 256, 0) and _geom(384) == (32, 256, 0)
    p, s, k = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    assert hb.lib().ssa_conv2d_wgrad_tile_geometry(720, 720, ctypes.byref(p), ctypes.byref(s), ctypes.byref(k)) != 0


def test_a_launch_of_ten_48_channel_layers_fits_one_round():
    # ten 48-channel layers at 256 x 256: 512 tiles each; strips of 8 give 640 workgroups = two and a half rounds of 256
    jobs = [_J(48, 256, 256) for _ in range(10)]
    assert _wgs(jobs, {}, 8) == 640
    fitted = hb._fit_tile_strips(jobs, 8)
    assert len(set(fitted.values())) == 1
    s = next(iter(fitted.values()))
    slots = _geom(48)[1]
    w = _wgs(jobs, fitted, 8)
    assert s > 8 and (w <= slots or w % slots == 0 or w % slots > slots * 0.8), (s, w)      # no nearly-empty tail round


def test_launches_are_fitted_separately_and_other_jobs_left_alone():
    G = hb._WGRAD_GROUP                                  # layers per grouped launch (csrc/group.h MAXJOBS)
    a = [_J(96, 128, 128) for _ in range(G)]            # one launch of the 96-channel instantiation
    b = [_J(96, 64, 64) for _ in range(4)]              # the next one: 4 x 5 x 2 = 40 at strips of 8
    odd = _J(48, 256, 256)
    odd.stride = 2                                       # not a halo-staged weight gradient: not planned
    fitted = hb._fit_tile_strips(a + b + [odd], 8)
    assert id(odd) not in fitted
    sa, sb = {fitted[id(j)] for j in a}, {fitted[id(j)] for j in b}
    assert len(sa) == 1 and len(sb) == 1
    assert sb == {8}                                     # 40 workgroups fit at the default length
    s = next(iter(sa))
    # no strip length in the planner's range does better (rounds x (strip + fixed cost))
    slots = _geom(96)[1]
    cost = lambda s_: -(-_wgs(a, {id(j): s_ for j in a}, 8) // slots) * (s_ + 2)      # noqa: E731
    assert cost(s) == min(cost(s_) for s_ in range(8, 8 * 8 + 1))
    if G > 16:
        assert s > 8                                     # twice the layers per launch: longer strips, fewer partials
