"""CPU: host-side logic that needs no GPU (engine option validation, transform layout, sharding)."""
import numpy as np
import pytest

from cilantro_amd import capi, distributed
from cilantro_amd.icp import CorrespondenceSearchDirection, CorrespondenceSearchHIP, _T_from_abi, _T_to_abi


def test_transform_abi_is_eigen_column_major():
    T = np.arange(16, dtype=np.float32).reshape(4, 4)
    a = _T_to_abi(T)
    # Eigen::Transform<float,3,Isometry>::data(): column-major -> element (r,c) at [c*4+r]
    assert a[1] == T[1, 0] and a[4] == T[0, 1] and a[12] == T[0, 3] and a[14] == T[2, 3]
    assert np.array_equal(_T_from_abi(a), T)


def test_engine_defaults_and_options():
    e = CorrespondenceSearchHIP(ctx=None)
    # correspondence_search_kd_tree.hpp:47-51 defaults
    assert e.getSearchDirection() == CorrespondenceSearchDirection.SECOND_TO_FIRST
    assert e.getMaxDistance() == np.float32(0.01 * 0.01) and e.getInlierFraction() == 1.0
    assert not e.getRequireReciprocality() and not e.getOneToOne()
    assert e.setMaxDistance(0.1 * 0.1) is e and e.getMaxDistance() == np.float32(0.1 * 0.1)
    # every knob of correspondence_search_kd_tree.hpp:239-271 is implemented
    assert e.setSearchDirection(CorrespondenceSearchDirection.BOTH).getSearchDirection() == CorrespondenceSearchDirection.BOTH
    assert e.setSearchDirection(CorrespondenceSearchDirection.FIRST_TO_SECOND) is e
    assert e.setRequireReciprocality(True).getRequireReciprocality()
    with pytest.raises(ValueError):
        e.setSearchDirection(7)
    assert e.setOneToOne(False) is e and e.setInlierFraction(1.0) is e
    assert e.setOneToOne(True).getOneToOne() and e.setInlierFraction(0.7).getInlierFraction() == 0.7   # filters are implemented


@pytest.mark.parametrize("n,world", [(10, 1), (10, 3), (7, 8), (1000001, 8), (0, 4)])
def test_shard_bounds_partition(n, world):
    parts = [distributed.shard_bounds(n, r, world) for r in range(world)]
    assert parts[0][0] == 0 and parts[-1][1] == n
    for (a, b), (c, d) in zip(parts, parts[1:]):
        assert b == c and b >= a
    sizes = [b - a for a, b in parts]
    assert max(sizes) - min(sizes) <= 1


def test_default_params_match_reference_defaults():
    p = distributed.default_params()
    assert p.metric == capi.METRIC_COMBINED and p.w_p2p == 0.0 and p.w_p2pl == 1.0      # combined_metric.hpp:44-47
    assert p.max_iter == 15 and abs(p.conv_tol - 1e-5) < 1e-12                           # icp_base.hpp:24-25
    assert p.max_opt_iter == 1 and abs(p.max_sq_dist - 1e-4) < 1e-10


def test_torch_default_stream_handle_maps_to_hip_stream_legacy():
    """The C ABI reserves NULL for "the context's own non-blocking stream".  torch reports handle 0 for its default
    stream: the Python layer must pass hipStreamLegacy ((hipStream_t)1) for it -- with NULL the sharded protocols'
    kernels were not ordered with torch / RCCL work at all.  None goes back to the own stream; other handles pass through."""
    from cilantro_amd.icp import Context

    seen = []

    class _Lib:
        @staticmethod
        def cilhip_set_stream(h, s):
            seen.append(s.value)
            return capi.OK

    c = object.__new__(Context)
    c._L, c._h, c._on_caller_stream = _Lib(), None, False
    c.set_stream(0)
    assert seen[-1] == 1 and c._on_caller_stream
    c.set_stream(0x7F00DEADBEE0)
    assert seen[-1] == 0x7F00DEADBEE0 and c._on_caller_stream
    c.set_stream(None)
    assert (seen[-1] or 0) == 0 and not c._on_caller_stream
    c._h = None   # nothing to destroy


def test_bench_forms_average_from_a_stratified_sample():
    """bench.py times a SAMPLE of the iterations (an event between dependent kernels idles the device).  The first iteration of a
    stretch of one form is not like the others and is always in the sample: a form's mean is formed per stratum (first of a
    stretch / the rest) and weighted by the strata's share of the RUN, so that the estimate equals the all-launch mean when the
    strata are homogeneous -- and is not pulled towards the first iteration by the sampling rate."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    forms = [1, 1] + [3] * 18
    per_it = [0.29, 0.35, 0.122] + [0.084] * 17
    timed = [(i, per_it[i]) for i in (0, 1, 2, 4, 8, 12, 16)]
    plain = {1: (0.64, 2), 3: (sum(per_it[i] for i in (2, 4, 8, 12, 16)), 5)}
    est = bench.stratified_form_timing(plain, forms, timed)
    assert abs(est[3][0] / est[3][1] - sum(per_it[2:]) / 18) < 1e-12            # = the mean over all 18 launches
    assert abs(plain[3][0] / plain[3][1] - 0.0916) < 1e-4                           # (the plain sample mean is 6 % off)
    assert abs(est[1][0] / est[1][1] - 0.32) < 1e-12 and est[3][1] == 5 and est[1][1] == 2
    # a second stretch of the same form after a fall back to the cold form: its first iteration is a stratum member too
    forms2 = [0, 3, 3, 3, 0, 3, 3, 3]
    timed2 = [(0, 0.5), (1, 0.3), (2, 0.1), (4, 0.5), (5, 0.3), (6, 0.1)]
    e2 = bench.stratified_form_timing({0: (1.0, 2), 3: (0.8, 4)}, forms2, timed2)
    assert abs(e2[3][0] / e2[3][1] - (0.3 * 2 + 0.1 * 4) / 6) < 1e-12
    # no trace (sharded runs) or nothing timed: the plain sums
    assert bench.stratified_form_timing(plain, [], timed) == plain and bench.stratified_form_timing(plain, forms, []) == plain
