"""GPU parity tests proper: the HIP kernels, called through the C ABI
(coda_neurips2023_amd.pointnet2._ext -> libcoda_hip.so), against the CPU oracle
on the same seeded inputs, against the golden fixtures generated from the
reference's Python layers, and at the full BASELINE sizes.

Bar: bit-exact for indices and pure gathers; fp32 sums that go through atomics
(order unspecified, as in the reference) within 1e-5 relative."""
import numpy as np
import pytest
import torch

from coda_neurips2023_amd.pointnet2 import _ext, pointnet2_utils
from coda_neurips2023_amd.synthetic_scenes import make_batch, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1, 2], ids=["dm0", "dm1", "dm2"])
def distance_mode(request, _distance_mode_default):
    """Every op test runs in all three distance-arithmetic modes (include/coda_pointnet2.h):
    the HIP kernels must be bit-exact against the oracle in the same mode.  Tests that use a
    golden fixture switch to the fixture's mode (conftest.golden_ops)."""
    from tests._modes import set_distance_mode
    set_distance_mode(request.param)
    return request.param


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def sample_centres(pc, fps_idx):
    return np.take_along_axis(pc, fps_idx[..., None].astype(np.int64).repeat(3, -1), 1)


# ---------------------------------------------------------------------------- FPS
@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 9, 2), (1, 2, 5), (3, 63, 17), (2, 64, 64),
                                   (2, 65, 30), (2, 300, 40), (1, 511, 64), (2, 512, 100),
                                   (2, 1024, 128), (2, 2048, 256), (1, 2049, 33), (2, 5000, 200),
                                   (1, 12345, 300)])
def test_fps_matches_oracle(dev, oracle, b, n, m):
    rng = np.random.default_rng(n * 31 + m)
    x = (rng.standard_normal((b, n, 3)) * 1.5).astype(np.float32)
    got = _ext.furthest_point_sampling(cu(x, dev), m).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (b, m)
    assert np.array_equal(got, oracle.furthest_point_sampling(x, m))


@pytest.mark.parametrize("n,m,kind", [(512, 512, "normal"), (513, 40, "dup"), (1024, 300, "grid"), (1025, 64, "normal"),
                                      (1500, 1500, "dup"), (2047, 256, "grid"), (2048, 128, "skip"), (2048, 2100, "dup")])
def test_fps_small_clouds(dev, oracle, n, m, kind):
    """512 .. 2048 points (the object queries' sampling, fps_t512_kernel with the cloud in LDS): the reference's tie
    rule (duplicates, lattice points with many equal distances), the skip rule and m > n."""
    rng = np.random.default_rng(n * 7 + m)
    if kind == "dup":
        base = (rng.random((37, 3), dtype=np.float32) * 3 + 1).astype(np.float32)
        x = base[rng.integers(0, 37, (3, n))]
    elif kind == "grid":
        x = rng.integers(1, 9, (3, n, 3)).astype(np.float32) * 0.25
    else:
        x = (rng.standard_normal((3, n, 3)) * 1.5).astype(np.float32)
        if kind == "skip":
            x[:, rng.integers(1, n, 200)] *= 1e-3   # inside the skip radius: never sampled
            x[1, 5:900] = 0.0
    got = _ext.furthest_point_sampling(cu(x, dev), m).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(x, m))


@pytest.mark.parametrize("n", [700, 1200, 4000, 20000])
def test_fps_tie_rule_with_duplicates(dev, oracle, n):
    rng = np.random.default_rng(n)
    base = (rng.random((50, 3), dtype=np.float32) * 3 + 1).astype(np.float32)
    x = base[rng.integers(0, 50, (2, n))]
    got = _ext.furthest_point_sampling(cu(x, dev), 80).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(x, 80))


@pytest.fixture(params=[16, 8], ids=["w16", "w8"])
def fps_waves(request):
    """Both workgroup shapes of the bucketed FPS kernels (16 waves x <= 20 slots, 8 waves x <= 40 slots)."""
    _ext.set_fps_waves(request.param)
    yield request.param
    _ext.set_fps_waves(0)


@pytest.mark.parametrize("n,m,kind", [(4096, 128, "normal"), (4097, 300, "dup"), (9000, 256, "plane"),
                                      (20000, 512, "dup"), (20480, 200, "normal"), (20000, 300, "skip"),
                                      (6000, 150, "same"), (20000, 2048, "line")])
def test_fps_bucketed_kernel_bit_exact(dev, oracle, fps_waves, n, m, kind):
    """n in [4096, 20480], m >= 128: the Morton-bucketed kernel with bounding-box pruning must
    select exactly the exhaustive scan's indices, including ties between duplicated points,
    degenerate extents and points inside the skip radius."""
    rng = np.random.default_rng(n + m)
    if kind == "normal":
        x = (rng.standard_normal((2, n, 3)) * 1.5).astype(np.float32)
    elif kind == "dup":  # 300 distinct locations -> massive ties
        base = (rng.random((300, 3), dtype=np.float32) * 4 - 2).astype(np.float32)
        x = base[rng.integers(0, 300, (2, n))]
    elif kind == "plane":  # zero extent along z, and along y for scene 1
        x = (rng.random((2, n, 3), dtype=np.float32) * 5).astype(np.float32)
        x[..., 2] = 1.25
        x[1, :, 1] = -0.5
    elif kind == "skip":  # a third of the points inside the skip radius, incl. point 0
        x = (rng.standard_normal((2, n, 3)) * 1.0).astype(np.float32)
        x[:, ::3] *= 0.01
    elif kind == "same":  # one location only
        x = np.tile(np.array([0.7, -1.1, 2.0], np.float32), (2, n, 1))
    else:  # points on a line, many exactly equal distances
        tline = np.round(rng.random((2, n, 1), dtype=np.float32) * 64) / 64
        x = (tline * np.array([1.0, 2.0, -0.5], np.float32)).astype(np.float32)
    got = _ext.furthest_point_sampling(cu(x, dev), m).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(x, m))


def test_fps_skip_rule(dev, oracle):
    x = np.zeros((2, 10, 3), np.float32)
    x[0, 3] = (0.03, 0, 0)
    x[0, 7] = (0.04, 0, 0)
    x[0, 5] = (np.sqrt(np.float32(1e-3)), 0, 0)
    # scene 1: every point within the skip radius -> all zeros
    got = _ext.furthest_point_sampling(cu(x, dev), 4).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(x, 4))
    assert not got[1].any()


def test_fps_short_scene_golden_and_synthetic(dev, oracle, golden_ops):
    for tag in ["small", "mid"]:
        xyz = golden_ops[f"{tag}_xyz"]
        m = golden_ops[f"{tag}_fps"].shape[1]
        got = _ext.furthest_point_sampling(cu(xyz, dev), m).cpu().numpy()
        assert np.array_equal(got, golden_ops[f"{tag}_fps"])
    pts = np.stack([make_scene(6000, seed=s, short_fraction=1.0) for s in (1, 2)])
    got = _ext.furthest_point_sampling(cu(pts, dev), 512).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(pts, 512))


@pytest.mark.parametrize("n,m", [(20481, 130), (40000, 2048), (40960, 128)])
def test_fps_two_workgroups_per_scene(dev, oracle, fps_waves, n, m):
    """20 480 < n <= 40 960 (ScanNet-sized clouds): two cooperating workgroups per scene, candidates exchanged
    through global mailboxes every round -- bit-exact like every other path."""
    pc, _, _ = make_batch(2, n, seed=n + 1)
    got = _ext.furthest_point_sampling(cu(pc, dev), m).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(pc, m))
    dup = pc.copy()
    dup[:, n // 2:] = dup[:, :n - n // 2]          # every point twice: ties between the two workgroups' halves
    got = _ext.furthest_point_sampling(cu(dup, dev), min(m, 300)).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(dup, min(m, 300)))


def test_fps_pair_that_loses_its_partner_is_loud(dev, oracle, distance_mode):
    """The two-workgroup kernel's one failure mode the reference's single block per scene (sampling_gpu.cu:72-176) does
    not have: the partner never answers.  Forced through the test hook (workgroup 1 of every pair exits at once, the
    poll limit lowered): the launch must END (no wait per round), the loss must reach the host (a word in pinned
    memory), the NEXT sampling call must refuse with a RuntimeError, and after that acknowledgement the operator must
    work -- and be bit-exact -- again."""
    import time
    from coda_neurips2023_amd import _lib
    if distance_mode != 1:
        pytest.skip("one arithmetic mode is enough for the failure path")
    lib = _lib.load()
    lib.coda_fps_lost_partner_events(1)
    pc, _, _ = make_batch(2, 40000, seed=77)
    x = cu(pc, dev)
    ref = oracle.furthest_point_sampling(pc, 300)
    assert np.array_equal(_ext.furthest_point_sampling(x, 300).cpu().numpy(), ref)
    assert lib.coda_fps_lost_partner_events(0) == 0
    _ext.check_sampling_status()
    t0 = time.perf_counter()
    wrong = _ext.furthest_point_sampling(x, 300, _dbg=(4000, 1))   # asynchronous: the call itself cannot know yet
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 2.0, "a lost partner must cost ONE bounded wait, not one per round"
    word = lib.coda_fps_lost_partner_events(0)
    assert word & 0x80000000 and (word & 0xffff) == 1, hex(word)   # given up in round 1
    w = wrong.cpu().numpy()
    assert w.min() >= 0 and w.max() < 40000                         # wrong, but never out of range
    assert not np.array_equal(w, ref)
    with pytest.raises(RuntimeError, match="lost its partner"):
        _ext.check_sampling_status()                                # acknowledges
    wrong = _ext.furthest_point_sampling(x, 300, _dbg=(4000, 1))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="lost its partner"):
        _ext.furthest_point_sampling(x, 300)                        # refuses to launch on top of the loss; acknowledges
    assert lib.coda_fps_lost_partner_events(0) == 0
    assert np.array_equal(_ext.furthest_point_sampling(x, 300).cpu().numpy(), ref)
    # a SLOW partner is not a lost one: with the default limit a chip-filling neighbour changes nothing
    assert lib.coda_fps_lost_partner_events(0) == 0


def test_fps_two_workgroups_under_cotenancy_stress(dev, oracle, distance_mode):
    """500 launches of the two-workgroup kernel at the ScanNet size (8 x 40 000 -> 2048) while another stream keeps
    the chip full (large GEMMs back to back, so the pair's workgroups are dispatched late and at different times),
    interleaved with launches on a second stream of its own: every result bit-equal to the oracle's, no lost-partner
    event.  (the pair's co-residency is not guaranteed by a plain launch; this is the evidence that a
    delayed partner is waited for, and test_fps_pair_that_loses_its_partner_is_loud that a missing one is loud.)"""
    import os
    from coda_neurips2023_amd import _lib
    if distance_mode != 1:
        pytest.skip("one arithmetic mode is enough for the stress loop")
    reps = int(os.environ.get("CODA_STRESS_REPS", "500"))
    lib = _lib.load()
    lib.coda_fps_lost_partner_events(1)
    pc, _, _ = make_batch(8, 40000, seed=4040)
    x = cu(pc, dev)
    ref = torch.from_numpy(oracle.furthest_point_sampling(pc, 2048)).to(dev)
    a = torch.randn(8192, 8192, device=dev)
    filler, second = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for i in range(reps):
        if i % 4 == 0:
            with torch.cuda.stream(filler):
                for _ in range(3):
                    a @ a                                         # ~10 ms of a full chip each
        if i % 5 == 4:                                            # two pairs of streams in flight at once
            second.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(second):
                got2 = _ext.furthest_point_sampling(x, 2048)
                bad += (got2 != ref).sum()
            torch.cuda.current_stream().wait_stream(second)
        got = _ext.furthest_point_sampling(x, 2048)
        bad += (got != ref).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0, f"{int(bad)} wrong indices in {reps} launches"
    assert lib.coda_fps_lost_partner_events(0) == 0


@pytest.mark.parametrize("n,m", [(30000, 300), (50000, 200), (30000, 100)])
def test_fps_streaming_paths(dev, oracle, n, m):
    """n > 24576 with few samples, or n > 40 960: running distances in LDS (<= ~40000) or in the workspace."""
    pc, _, _ = make_batch(2, n, seed=n)
    got = _ext.furthest_point_sampling(cu(pc, dev), m).cpu().numpy()
    assert np.array_equal(got, oracle.furthest_point_sampling(pc, m))


def test_fps_full_size_bit_exact(dev, oracle, fps_waves):
    """BASELINE shape: B=8, N=20000 -> 2048, then 2048 -> 256 (query sampling)."""
    pc, _, _ = make_batch(8, 20000, seed=1234)
    got = _ext.furthest_point_sampling(cu(pc, dev), 2048).cpu().numpy()
    ref = oracle.furthest_point_sampling(pc, 2048)
    assert np.array_equal(got, ref)
    assert all(len(np.unique(r)) == 2048 for r in got)  # no duplicates in these scenes
    enc_xyz = sample_centres(pc, got)
    got2 = _ext.furthest_point_sampling(cu(enc_xyz, dev), 256).cpu().numpy()
    assert np.array_equal(got2, oracle.furthest_point_sampling(enc_xyz, 256))


# --------------------------------------------------------------------- ball query
@pytest.mark.parametrize("b,n,m,r,s", [(1, 1, 1, 0.5, 1), (2, 9, 2, 5.0, 3), (2, 9, 2, 10.0, 6),
                                       (2, 70, 5, 0.8, 4), (2, 1000, 37, 0.3, 16),
                                       (2, 3000, 130, 0.25, 64), (1, 5000, 64, 0.4, 32),
                                       (1, 2000, 40, 0.5, 200), (1, 600, 9, 10.0, 700),
                                       (1, 300, 5, 10.0, 3000)])
def test_ball_query_matches_oracle(dev, oracle, b, n, m, r, s):
    if n >= 600:
        pc, _, _ = make_batch(b, n, seed=n + s)
    else:
        pc = (np.random.default_rng(n).standard_normal((b, n, 3))).astype(np.float32)
    new = pc[:, np.random.default_rng(m).integers(0, n, m)].copy()
    new[:, 0] += 100.0  # an empty ball per scene -> all-zero row
    got = _ext.ball_query(cu(new, dev), cu(pc, dev), r, s).cpu().numpy()
    ref = oracle.ball_query(new, pc, r, s)
    assert got.dtype == np.int32 and np.array_equal(got, ref)
    assert not got[:, 0].any()


@pytest.mark.parametrize("algorithm", ["scan", "grid"])
@pytest.mark.parametrize("n,m,r,s", [(1024, 100, 0.3, 16), (3000, 130, 0.25, 64), (20000, 500, 0.2, 64),
                                     (20000, 300, 0.4, 32), (5000, 64, 1.5, 128), (40000, 256, 0.2, 64),
                                     (20001, 2048, 0.2, 64), (12345, 1000, 0.05, 8),
                                     # rows shorter than one 32-sample pass of the eight-lane query kernel's vectorised
                                     # output (the channels-last store is cooperative across the 8 lanes)
                                     (4096, 257, 0.3, 4), (4096, 257, 0.3, 24), (20000, 300, 0.3, 48),
                                     (4096, 64, 0.3, 6), (4096, 64, 0.3, 63)])
def test_ball_query_scan_and_grid_paths(dev, oracle, algorithm, n, m, r, s):
    pc, _, _ = make_batch(2, n, seed=n + m)
    rng = np.random.default_rng(s)
    new = pc[:, rng.integers(0, n, m)].copy()
    new[:, 0] += 100.0                      # far outside the cloud: empty ball
    new[:, 1] = pc.min(1) - 0.05            # just outside the bounding box corner
    new[:, 2] += rng.normal(0, 0.05, 3).astype(np.float32)   # not a cloud point
    ref = oracle.ball_query(new, pc, r, s)
    d_new, d_pc = cu(new, dev), cu(pc, dev)
    got = _ext.ball_query(d_new, d_pc, r, s, algorithm=algorithm)
    assert np.array_equal(got.cpu().numpy(), ref)
    idx, grouped = _ext.query_and_group_xyz(d_new, d_pc, r, s, True, algorithm=algorithm)
    assert np.array_equal(idx.cpu().numpy(), ref)
    exp = np.take_along_axis(pc, ref.reshape(2, -1, 1).astype(np.int64).repeat(3, -1), 1).reshape(2, m, s, 3)
    exp_raw = exp - new[:, :, None, :]
    exp = exp_raw * (np.float32(1.0) / np.float32(r))
    np.testing.assert_array_equal(grouped.cpu().numpy(), exp.transpose(0, 3, 1, 2))
    # channels-last (B,M,S,3) rows -- what the fused set-abstraction front consumes -- on poisoned memory
    # (torch.empty hands back recycled blocks: a piece the kernel skips would otherwise often LOOK right)
    poison = [torch.full((2, m, s), -7, dtype=torch.int32, device=dev), torch.full((2, m, s, 3), float("nan"), device=dev)]
    del poison
    idx_cl, grouped_cl = _ext.query_and_group_xyz(d_new, d_pc, r, s, True, algorithm=algorithm, channels_last=True)
    assert np.array_equal(idx_cl.cpu().numpy(), ref)
    np.testing.assert_array_equal(grouped_cl.cpu().numpy(), exp)
    _, raw_cl = _ext.query_and_group_xyz(d_new, d_pc, r, s, False, algorithm=algorithm, channels_last=True)
    np.testing.assert_array_equal(raw_cl.cpu().numpy(), exp_raw)


def test_ball_query_grid_stress(dev, oracle):
    """Balls holding far more hits than the LDS hit buffer (rank-and-keep overflow path),
    duplicate points, non-finite points, a far outlier that stretches the grid."""
    rng = np.random.default_rng(0)
    n = 6000
    pc = (rng.standard_normal((2, n, 3)) * 0.15).astype(np.float32)       # ~all within r of the origin
    pc[:, 100:200] = pc[:, 0:100]                                           # exact duplicates
    pc[0, 7] = (np.nan, 0, 0)
    pc[0, 8] = (np.inf, 0, 0)
    pc[1, 9] = (1e6, -1e6, 3e5)                                             # outlier
    new = np.concatenate([np.zeros((2, 1, 3), np.float32), pc[:, 1000:1063]], 1)
    for r, s in [(0.5, 64), (0.2, 16), (2.0, 128)]:
        ref = oracle.ball_query(new, pc, r, s)
        for algorithm in ("grid", "scan"):
            got = _ext.ball_query(cu(new, dev), cu(pc, dev), r, s, algorithm=algorithm)
            assert np.array_equal(got.cpu().numpy(), ref), (r, s, algorithm)


def test_ball_query_golden(dev, golden_ops):
    for tag in ["small", "mid"]:
        got = _ext.ball_query(cu(golden_ops[f"{tag}_new_xyz"], dev), cu(golden_ops[f"{tag}_xyz"], dev),
                              float(golden_ops[f"{tag}_radius"]), int(golden_ops[f"{tag}_nsample"]))
        assert np.array_equal(got.cpu().numpy(), golden_ops[f"{tag}_ball_idx"])


def test_ball_query_and_fused_group_full_size(dev, oracle):
    """BASELINE shape: B=8, N=20000, M=2048, r=0.2, nsample=64 + fused grouping."""
    pc, _, _ = make_batch(8, 20000, seed=1234)
    fps = oracle.furthest_point_sampling(pc, 2048)
    new = sample_centres(pc, fps)
    d_pc, d_new = cu(pc, dev), cu(new, dev)
    ref = oracle.ball_query(new, pc, 0.2, 64)
    got = _ext.ball_query(d_new, d_pc, 0.2, 64)
    assert np.array_equal(got.cpu().numpy(), ref)
    idx, grouped = _ext.query_and_group_xyz(d_new, d_pc, 0.2, 64, True)
    assert np.array_equal(idx.cpu().numpy(), ref)
    # properties: rows ascending up to the pad, the centre itself is always a member
    r = idx.cpu().numpy()
    assert (r[..., 0][..., None] <= r).all()
    assert ((r == fps[..., None]).any(-1) | (np.diff(r, axis=-1) >= 0).all(-1)).all()
    # grouped = (xyz[idx] - centre) * (1/0.2f) in fp32 (torch's GPU div-by-scalar)
    exp = (np.take_along_axis(pc[:, None], r.reshape(8, -1, 1).astype(np.int64).repeat(3, -1)[:, None], 2)
           .reshape(8, 2048, 64, 3) - new[:, :, None, :]) * (np.float32(1.0) / np.float32(0.2))
    np.testing.assert_array_equal(grouped.cpu().numpy(), exp.transpose(0, 3, 1, 2))
    assert np.abs(grouped.cpu().numpy()).max() < 1.0 + 1e-5  # inside the unit ball


def test_fused_group_equals_unfused_sequence(dev):
    pc, _, _ = make_batch(2, 4000, seed=3)
    d_pc = cu(pc, dev)
    d_new = d_pc[:, :300].contiguous()
    for normalize in (False, True):
        idx, grouped = _ext.query_and_group_xyz(d_new, d_pc, 0.3, 32, normalize)
        idx2 = _ext.ball_query(d_new, d_pc, 0.3, 32)
        g2 = _ext.group_points(d_pc.transpose(1, 2).contiguous(), idx2)
        g2 = g2 - d_new.transpose(1, 2).unsqueeze(-1)
        if normalize:
            g2 = g2 * (1.0 / torch.tensor(0.3, dtype=torch.float32)).item()
        assert torch.equal(idx, idx2)
        torch.testing.assert_close(grouped, g2, rtol=2e-7, atol=0)


# ------------------------------------------------------------ gathers / scatter-adds
@pytest.mark.parametrize("b,c,n,m,s", [(1, 1, 1, 1, 1), (2, 3, 50, 7, 5), (2, 6, 1024, 128, 64),
                                       (1, 19, 333, 21, 9), (2, 256, 2048, 64, 32)])
def test_group_and_gather_exact(dev, oracle, b, c, n, m, s):
    rng = np.random.default_rng(c * 7 + n)
    pts = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, s)).astype(np.int32)
    got = _ext.group_points(cu(pts, dev), cu(idx, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.group_points(pts, idx))
    idx1 = rng.integers(0, n, (b, m)).astype(np.int32)
    got = _ext.gather_points(cu(pts, dev), cu(idx1, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.gather_points(pts, idx1))


@pytest.mark.parametrize("b,c,n,m,s", [(2, 3, 50, 7, 5), (2, 6, 1024, 128, 64), (1, 19, 333, 21, 9)])
def test_scatter_add_grads(dev, oracle, b, c, n, m, s):
    rng = np.random.default_rng(c + n)
    idx = rng.integers(0, n, (b, m, s)).astype(np.int32)
    go = rng.standard_normal((b, c, m, s)).astype(np.float32)
    got = _ext.group_points_grad(cu(go, dev), cu(idx, dev), n).cpu().numpy()
    np.testing.assert_allclose(got, oracle.group_points_grad(go, idx, n), rtol=1e-5, atol=1e-5)
    idx1 = rng.integers(0, n, (b, m)).astype(np.int32)
    go1 = rng.standard_normal((b, c, m)).astype(np.float32)
    got = _ext.gather_points_grad(cu(go1, dev), cu(idx1, dev), n).cpu().numpy()
    np.testing.assert_allclose(got, oracle.gather_points_grad(go1, idx1, n), rtol=1e-5, atol=1e-5)
    # unique indices -> no atomics collisions -> bit exact
    perm = np.stack([rng.permutation(n)[:m] for _ in range(b)]).astype(np.int32)
    got = _ext.gather_points_grad(cu(go1, dev), cu(perm, dev), n).cpu().numpy()
    assert np.array_equal(got, oracle.gather_points_grad(go1, perm, n))


def test_scatter_add_grads_are_deterministic_and_accurate(dev, distance_mode):
    """The default adjoints of gather / group accumulate in 64-bit fixed point (include/coda_pointnet2.h): the same bits
    from run to run under heavy collisions (the masked encoder's shape, 32 768 entries onto 2048 targets per scene and
    channel; and the worst case, every entry onto ONE target), closer to the float64 sum than the float atomics are,
    and NaN in the (scene, channel) row of a non-finite input.  (The reference's atomicAdd form, CODA_SCATTER=atomic, is neither.)"""
    if distance_mode != 1:
        pytest.skip("no distance arithmetic in this operator")
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    b, c, n, m, s = 2, 64, 2048, 1024, 32
    idx = rng.integers(0, n, (b, m, s)).astype(np.int32)
    idx[1] = 7                                                   # scene 1: every entry collides on target 7
    go = (rng.standard_normal((b, c, m, s)) * np.exp(rng.uniform(-8, 8, (b, c, 1, 1)))).astype(np.float32)
    d_go, d_idx = cu(go, dev), cu(idx, dev)
    first = _ext.group_points_grad(d_go, d_idx, n)
    for _ in range(5):
        assert torch.equal(_ext.group_points_grad(d_go, d_idx, n), first), "not reproducible"
    exact = np.zeros((b, c, n), np.float64)
    for bi in range(b):
        np.add.at(exact[bi].T, idx[bi].reshape(-1), go[bi].reshape(c, -1).T.astype(np.float64))
    got = first.cpu().numpy().astype(np.float64)
    scale = np.abs(go).max()
    err_det = np.abs(got - exact).max() / scale
    atomic = torch.empty_like(first)
    st = lib.coda_group_points_grad_f32(d_go.data_ptr(), d_idx.data_ptr(), atomic.data_ptr(), b, c, n, m, s,
                                        _lib.current_stream_handle())
    assert st == 0
    err_atomic = np.abs(atomic.cpu().numpy().astype(np.float64) - exact).max() / scale
    print(f"max error / max |g|: fixed point {err_det:.2e}, float atomics {err_atomic:.2e}")
    assert err_det <= 2.0 ** -23 * 1.01 * np.abs(exact).max() / scale + 1e-12   # one float32 rounding of the result
    assert err_det <= err_atomic + 1e-12
    # the fixed-point scale is per (scene, channel) row (round 6): a channel 7 decades below the largest one is as
    # accurate, relative to ITSELF, as the largest (with one scale for the tensor its addends rounded to zero)
    row_err = np.abs(got - exact).max(-1) / (np.abs(exact).max(-1) + 1e-300)
    assert row_err.max() <= 2.0 ** -23 * 1.01, row_err.max()
    # gather_points_grad: the same code with one entry per row
    idx1 = rng.integers(0, 16, (b, m)).astype(np.int32)
    go1 = rng.standard_normal((b, c, m)).astype(np.float32)
    g1 = _ext.gather_points_grad(cu(go1, dev), cu(idx1, dev), n)
    assert torch.equal(_ext.gather_points_grad(cu(go1, dev), cu(idx1, dev), n), g1)
    ref1 = np.zeros((b, c, n), np.float64)
    for bi in range(b):
        np.add.at(ref1[bi].T, idx1[bi], go1[bi].T.astype(np.float64))
    np.testing.assert_allclose(g1.cpu().numpy(), ref1, rtol=2e-7, atol=1e-7)
    # a non-finite gradient: NaN in its (scene, channel) row (stated in the header), the other rows untouched
    bad = d_go.clone()
    bad[0, 3, 5, 1] = float("inf")
    res = _ext.group_points_grad(bad, d_idx, n)
    assert torch.isnan(res[0, 3]).all()
    keep = torch.ones(b, c, dtype=torch.bool, device=dev)
    keep[0, 3] = False
    assert torch.equal(res[keep], first[keep])
    assert not _ext.group_points_grad(torch.zeros_like(d_go), d_idx, n).any()
    # too small a workspace is refused
    out = torch.empty((b, c, n), device=dev)
    ws = torch.empty(64, dtype=torch.int64, device=dev)
    assert lib.coda_group_points_grad_det_f32(d_go.data_ptr(), d_idx.data_ptr(), out.data_ptr(), b, c, n, m, s, ws.data_ptr(),
                                              ws.numel() * 8, _lib.current_stream_handle()) == _lib.CODA_ENOSPC


# ----------------------------------------------------------- three_nn / interpolate
@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 50, 2), (2, 50, 30), (2, 1024, 128), (1, 3000, 1500)])
def test_three_nn_matches_oracle(dev, oracle, b, n, m):
    rng = np.random.default_rng(n + m)
    u = rng.random((b, n, 3), dtype=np.float32)
    k = rng.random((b, m, 3), dtype=np.float32)
    if m > 5:
        k[:, 5] = k[:, 2]  # tie -> lower index first
    d2, idx = _ext.three_nn(cu(u, dev), cu(k, dev))
    rd2, ridx = oracle.three_nn(u, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)  # inf where m < 3


def test_three_interpolate_known_answer_and_golden(dev, oracle, golden_ops):
    g = golden_ops
    out = _ext.three_interpolate(cu(g["kat_feats"], dev), cu(g["kat_idx"], dev), cu(g["kat_weight"], dev))
    assert np.array_equal(out.cpu().numpy(), g["kat_interp"])
    grad = _ext.three_interpolate_grad(torch.ones_like(out), cu(g["kat_idx"], dev), cu(g["kat_weight"], dev), 4)
    assert np.array_equal(grad.cpu().numpy(), g["kat_grad"])
    for tag in ["small", "mid"]:
        out = _ext.three_interpolate(cu(g[f"{tag}_known_feats"], dev), cu(g[f"{tag}_nn_idx"], dev),
                                     cu(g[f"{tag}_nn_weight"], dev))
        assert np.array_equal(out.cpu().numpy(), g[f"{tag}_interp"])
        m = g[f"{tag}_known_feats"].shape[2]
        grad = _ext.three_interpolate_grad(cu(g[f"{tag}_interp_gw"], dev), cu(g[f"{tag}_nn_idx"], dev),
                                           cu(g[f"{tag}_nn_weight"], dev), m)
        np.testing.assert_allclose(grad.cpu().numpy(), g[f"{tag}_interp_grad"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------- autograd functions vs the reference
def test_autograd_functions_against_reference_golden(dev, golden_ops):
    g = golden_ops
    for tag in ["small", "mid"]:
        xyz = cu(g[f"{tag}_xyz"], dev)
        feats = cu(g[f"{tag}_feats"], dev).requires_grad_(True)
        m = g[f"{tag}_fps"].shape[1]
        inds = pointnet2_utils.furthest_point_sample(xyz, m)
        assert np.array_equal(inds.cpu().numpy(), g[f"{tag}_fps"])
        new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        assert np.array_equal(new_xyz.cpu().numpy(), g[f"{tag}_new_xyz"])
        idx = pointnet2_utils.ball_query(float(g[f"{tag}_radius"]), int(g[f"{tag}_nsample"]), xyz, new_xyz)
        grouped = pointnet2_utils.grouping_operation(feats, idx)
        assert np.array_equal(grouped.detach().cpu().numpy(), g[f"{tag}_grouped"])
        (grouped * cu(g[f"{tag}_group_gw"], dev)).sum().backward()
        np.testing.assert_allclose(feats.grad.cpu().numpy(), g[f"{tag}_group_grad"], rtol=1e-4, atol=1e-5)
        feats.grad = None
        gathered = pointnet2_utils.gather_operation(feats, inds)
        assert np.array_equal(gathered.detach().cpu().numpy(), g[f"{tag}_gathered"])
        (gathered * cu(g[f"{tag}_gather_gw"], dev)).sum().backward()
        np.testing.assert_allclose(feats.grad.cpu().numpy(), g[f"{tag}_gather_grad"], rtol=1e-4, atol=1e-5)
        dist, nn_idx = pointnet2_utils.three_nn(xyz, new_xyz)
        assert np.array_equal(nn_idx.cpu().numpy(), g[f"{tag}_nn_idx"])
        np.testing.assert_allclose(dist.cpu().numpy(), g[f"{tag}_nn_dist"], rtol=2.4e-7)
        assert not inds.requires_grad and not idx.requires_grad


def test_runs_on_non_default_stream(dev, oracle):
    pc, _, _ = make_batch(2, 3000, seed=21)
    d_pc = cu(pc, dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        got = _ext.furthest_point_sampling(d_pc, 128)
    s.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle.furthest_point_sampling(pc, 128))
