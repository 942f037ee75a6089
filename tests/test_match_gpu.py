"""K7 parity: HIP Hamming kNN-2 (through the C ABI) vs the oracle -- bit-exact indices and distances."""
import numpy as np
import pytest

from monocular_slam_amd import api, synth

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, q, t):
    idx, dist = ctx.match_knn2(q, t)
    rc, oidx, odist = oracle.match_knn2(q, t)
    assert rc == 0
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    return idx, dist


def test_c1m_2000x2000(ctx, oracle):
    a, b, perm = synth.descriptor_sets(2000, 2003)
    idx, dist = _check(ctx, oracle, b, a)
    # ratio test: same survivors as the oracle, and most true matches recovered
    for ratio in (0.8, 0.85):
        qi, ti, d = api.ratio_filter(idx, dist, ratio)
        oq, ot, od = oracle.ratio_filter(idx, dist, ratio)
        np.testing.assert_array_equal(qi, oq)
        np.testing.assert_array_equal(ti, ot)
        np.testing.assert_array_equal(d, od)
    assert (perm[qi] == ti).mean() > 0.99


@pytest.mark.parametrize("nq,nt", [(1, 2), (2, 2), (63, 65), (64, 64), (257, 255), (513, 700), (500, 3), (3, 1500), (4096, 4096)])
def test_ragged_sizes(ctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)
    _check(ctx, oracle, q, t)


@pytest.mark.parametrize("nq,nt", [(31, 33), (32, 32), (33, 31), (127, 129), (128, 128), (129, 96), (2000, 95), (95, 2000), (1999, 2017), (65, 4095), (9000, 10000), (300, 16128), (300, 16200)])
def test_sizes_around_the_mfma_tiles(ctx, oracle, nq, nt):
    """The matcher works on tiles of 32 train rows x 64 query columns per wavefront and cuts the train tiles into shares
    (hamming_mfma_k): sizes one off the tile, wavefront and workgroup boundaries, shares with no tile at all (nt = 95 with 4
    shares of 1 tile), rows of all-zero / all-one bits (dot products at both ends of the int8 range), and the largest train set
    whose indices fit beside the distance in the MFMA's keys (16128 rows) next to one that takes the VALU form (16200)."""
    rng = np.random.default_rng(nq * 104729 + nt)
    q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)
    q[0] = 0; q[-1] = 255
    t[nt // 2] = 0; t[-1] = 255; t[0] = q[nq // 2]          # distances 0 and 256 occur
    _check(ctx, oracle, q, t)


def test_ties_go_to_lower_train_index(ctx, oracle):
    """Collisions: duplicated train rows and all-equal distances must rank by train index."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=(40, 32), dtype=np.uint8)
    t = np.concatenate([base, base, base[::-1]])       # every row appears 3 times
    q = np.concatenate([base, np.zeros((5, 32), np.uint8), np.full((5, 32), 255, np.uint8)])
    idx, dist = _check(ctx, oracle, q, t)
    assert (dist[:40, 0] == 0).all() and (dist[:40, 1] == 0).all()
    assert (idx[:40, 0] == np.arange(40)).all() and (idx[:40, 1] == np.arange(40) + 40).all()
    # constant train set: best = row 0, second = row 1
    t2 = np.zeros((300, 32), np.uint8)
    idx2, dist2 = _check(ctx, oracle, q, t2)
    assert (idx2[:, 0] == 0).all() and (idx2[:, 1] == 1).all()


def test_too_few_train_rows(ctx):
    q = np.zeros((4, 32), np.uint8)
    with pytest.raises(api.MslamError) as e:
        ctx.match_knn2(q, np.zeros((1, 32), np.uint8))
    assert e.value.status == api.ERR_TOO_FEW
    idx, dist = ctx.match_knn2(np.zeros((0, 32), np.uint8), np.zeros((5, 32), np.uint8))
    assert idx.shape == (0, 2)


def test_batched_device_path_with_ratio(ctx, oracle):
    """Batched device entry point: ragged per-problem counts, fused ratio test and ordered compaction."""
    B, cap = 5, 700
    rng = np.random.default_rng(11)
    nq = np.array([700, 1, 333, 0, 512], np.int32)
    nt = np.array([650, 700, 2, 100, 1], np.int32)   # last problem has < 2 train rows
    q = rng.integers(0, 256, size=(B, cap, 32), dtype=np.uint8)
    t = rng.integers(0, 256, size=(B, cap, 32), dtype=np.uint8)
    # plant near-duplicates so that the ratio test keeps something
    for b in range(B):
        k = min(nq[b], nt[b]) // 2
        t[b, :k] = q[b, :k]
        t[b, :k, 0] ^= 1
    D = api.DevBuf.from_array
    dq, dt, dnq, dnt = D(ctx, q), D(ctx, t), D(ctx, nq), D(ctx, nt)
    didx = D(ctx, np.full((B, cap, 2), -7, np.int32))
    ddist = api.DevBuf(ctx, B * cap * 2 * 4)
    mq, mt, md = api.DevBuf(ctx, B * cap * 4), api.DevBuf(ctx, B * cap * 4), api.DevBuf(ctx, B * cap * 4)
    mc = api.DevBuf(ctx, B * 4)
    ctx.match_knn2_batch_dev(dq.ptr, dnq.ptr, dt.ptr, dnt.ptr, cap, B, didx.ptr, ddist.ptr, 0.8, mq.ptr, mt.ptr, md.ptr, mc.ptr)
    ctx.sync()
    idx, dist = didx.download(np.int32, (B, cap, 2)), ddist.download(np.int32, (B, cap, 2))
    mcount = mc.download(np.int32, (B,))
    hq, ht, hd = mq.download(np.int32, (B, cap)), mt.download(np.int32, (B, cap)), md.download(np.float32, (B, cap))
    for b in range(B):
        if nt[b] < 2:
            assert mcount[b] == 0 and (idx[b, : nq[b]] == -1).all()
            continue
        rc, oidx, odist = oracle.match_knn2(q[b, : nq[b]], t[b, : nt[b]])
        np.testing.assert_array_equal(idx[b, : nq[b]], oidx)
        np.testing.assert_array_equal(dist[b, : nq[b]], odist)
        oq, ot, od = oracle.ratio_filter(oidx, odist, 0.8)
        assert mcount[b] == len(oq)
        np.testing.assert_array_equal(hq[b, : mcount[b]], oq)
        np.testing.assert_array_equal(ht[b, : mcount[b]], ot)
        np.testing.assert_array_equal(hd[b, : mcount[b]], od)


def test_valu_matcher_kept_for_comparison_still_matches_the_oracle():
    """MSLAM_MATCH_MFMA=0 selects the xor/popcount kernel (the switch is read once per process, so this runs in a child)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np\n"
        "from monocular_slam_amd import api\n"
        "from oracle import binding as ob\n"
        "ctx = api.Context(0, 640, 480, 4096, 1)\n"
        "rng = np.random.default_rng(3)\n"
        "for nq, nt in ((2000, 2000), (257, 255), (33, 700), (1, 2)):\n"
        "    q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8); t = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)\n"
        "    t[:min(nq, nt) // 2] = q[:min(nq, nt) // 2]\n"
        "    idx, dist = ctx.match_knn2(q, t); rc, oi, od = ob.match_knn2(q, t)\n"
        "    assert rc == 0 and (idx == oi).all() and (dist == od).all(), (nq, nt)\n"
        "ctx.close(); print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, MSLAM_MATCH_MFMA="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
