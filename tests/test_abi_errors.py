"""Error behaviour of the C ABI (include/strotss_hip.h: STROTSS_EINVAL / _EALIGN / _ERANGE): every entry point validates its
arguments BEFORE its first launch and reports through its int status -- never a fault, never a silent no-op.  No GPU needed:
the pointers below are never dereferenced because the calls are refused first (a call that got past its checks would fail
differently -- hipErrorNoDevice here, a launch on the GPU box -- and the asserted code would not match)."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

EINVAL, EALIGN, ERANGE = -1, -2, -3
P = C.c_void_p(0x10000)          # "some buffer": non-null, never touched
NULL = None
WS = 1 << 30


@pytest.fixture(scope="module")
def lib():
    from nn import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _hip.load_library()


def test_convolutions_refuse_bad_shapes_and_missing_buffers(lib):
    # generic layer: null input; cin not a multiple of 32; cout not a multiple of 64
    assert lib.strotss_conv3x3_relu_fwd(NULL, 8, 8, 64, P, P, 64, P, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_relu_fwd(P, 8, 8, 33, P, P, 64, P, P, WS, NULL) == EALIGN
    assert lib.strotss_conv3x3_relu_fwd(P, 8, 8, 64, P, P, 96, P, P, WS, NULL) == EALIGN
    assert lib.strotss_conv3x3_relu_fwd(P, 0, 8, 64, P, P, 64, P, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_dgrad(P, 8, 8, 64, P, 65, NULL, P, 0, P, WS, NULL) == EALIGN
    # pooled finish (ABI 8): needs the pooled output and a workspace; the gradient in front of the pool must be 2x the map
    assert lib.strotss_conv3x3_relu_pool_fwd(P, 8, 8, 64, P, P, 64, P, NULL, NULL, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_relu_pool_fwd(P, 8, 8, 64, P, P, 64, P, P, NULL, NULL, 0, NULL) == EINVAL
    assert lib.strotss_conv3x3_dgrad_unpool(P, 4, 4, 64, P, 64, P, P, 9, 12, 0, P, WS, NULL) == EINVAL       # 12 / 2 != 4
    assert lib.strotss_conv3x3_dgrad_unpool(P, 4, 4, 64, P, 64, NULL, P, 8, 8, 0, P, WS, NULL) == EINVAL     # no argmax codes
    # Winograd forms: tile size 2 or 4 only; sign words and accumulate are F(4x4,3x3) features; accumulate needs a mask source
    assert lib.strotss_conv3x3_winograd_dgrad(P, 8, 8, 64, P, NULL, NULL, 64, 3, P, NULL, P, 0, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_winograd_dgrad(P, 8, 8, 64, P, NULL, NULL, 64, 2, P, P, P, 0, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_winograd_dgrad(P, 8, 8, 64, P, NULL, NULL, 64, 4, NULL, NULL, P, 1, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_winograd_dgrad(P, 8, 8, 64, P, NULL, NULL, 64, 2, P, NULL, P, 1, P, WS, NULL) == EINVAL
    assert lib.strotss_conv3x3_winograd_dgrad(P, 8, 8, 48, P, NULL, NULL, 64, 4, P, NULL, P, 0, P, WS, NULL) == EALIGN
    assert lib.strotss_conv3x3_winograd_route(8, 8, 64, 64, 3, 1, 1) == EINVAL
    assert lib.strotss_conv3x3_winograd_route(0, 8, 64, 64, 4, 1, 1) == EINVAL
    assert lib.strotss_conv3x3_winograd_route(8, 8, 64, 64, 4, 1, 1) >= 0


def test_losses_refuse_unpadded_rows_and_oversized_lists(lib):
    f = C.c_float
    args = lambda n, d, ld, ns: (P, P, n, d, ld, P, P, P, ns, P, P, f(1), f(1), f(1), f(1), P, P, P, P, P, P, WS, NULL)
    assert lib.strotss_step_losses_fwd_bwd(*args(1024, 2179, 2180, 1024)) == EALIGN          # ld % 32 != 0
    assert lib.strotss_step_losses_fwd_bwd(*args(1024, 2179, 2176, 1024)) == EINVAL          # ld < d
    assert lib.strotss_step_losses_fwd_bwd(*args(1024, 2179, 2208, 4096)) == ERANGE          # style rows > the tie-list capacity
    assert lib.strotss_step_losses_fwd_bwd(*args(0, 2179, 2208, 1024)) == EINVAL
    bad = list(args(1024, 2179, 2208, 1024)); bad[15] = NULL                                   # no gradient buffer
    assert lib.strotss_step_losses_fwd_bwd(*bad) == EINVAL


def test_sampling_entries_refuse_bad_descriptors(lib):
    from nn import _hip
    d = _hip.DrawT()
    assert lib.strotss_index_draw(C.byref(d), NULL) == EINVAL                                  # all zero
    d.h, d.w, d.step_x, d.step_y, d.sample_size, d.n_regions = 64, 64, 1, 1, 1024, 1
    d.counter = 0x10000
    assert lib.strotss_index_draw(C.byref(d), NULL) == EINVAL                                  # region 0 has no output buffer
    d.idx[0] = 0x10000
    d.n_regions = _hip.MAX_DRAW_REGIONS + 1
    assert lib.strotss_index_draw(C.byref(d), NULL) == EINVAL
    d.n_regions, d.sample_size = 1, 4096
    assert lib.strotss_index_draw(C.byref(d), NULL) == EINVAL                                  # more samples than the kernel's threads
    d.sample_size, d.h, d.w = 1024, 1024, 1024
    assert lib.strotss_index_draw(C.byref(d), NULL) == ERANGE                                  # 2^20 candidates at step 1
    assert lib.strotss_index_draw_max_candidates(1024, 1024, 8, 8) == 128 * 128
    assert lib.strotss_index_draw_max_candidates(0, 4, 1, 1) == EINVAL
    m = _hip.MapsT()
    m.n_maps = 1
    m.h[0], m.w[0], m.c[0] = 8, 8, 64
    m.map[0] = 0x10000
    assert lib.strotss_hypercol_scatter(C.byref(m), P, 16, P, 64, 1, 0, 1, NULL) == EINVAL     # no gradient buffer for map 0
    m.gmap[0] = 0x10000
    assert lib.strotss_hypercol_scatter(C.byref(m), P, 16, P, 64, 1, 1, 1, NULL) == ERANGE     # empty map range
    assert lib.strotss_hypercol_scatter(C.byref(m), P, 16, P, 32, 1, 0, 1, NULL) == EINVAL     # rows narrower than the maps
    assert lib.strotss_hypercol_scatter(C.byref(m), NULL, 16, P, 64, 1, 0, 1, NULL) == EINVAL
    assert lib.strotss_hypercol_scatter_plan(C.byref(m), P, 2048, P, 1 << 20, NULL) == ERANGE  # 4 taps x 2048 samples > the plan
