"""K1-K6 parity: HIP ORB (through the C ABI) vs the oracle -- bit-exact keypoints and descriptors."""
import numpy as np
import pytest

from monocular_slam_amd import api, synth

pytestmark = pytest.mark.gpu


def _same_keypoints(kg, dg, ko, do):
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        np.testing.assert_array_equal(kg[f], ko[f], err_msg=f)
    np.testing.assert_array_equal(dg, do)


def _oracle_params(oracle, p):
    op = oracle.orb_default_params(p.nfeatures)
    for f, _ in api.OrbParams._fields_:
        setattr(op, f, getattr(p, f))
    return op


@pytest.mark.parametrize("w,h,nf,seed", [(640, 480, 500, 1001), (1920, 1080, 2000, 2002), (333, 257, 300, 7), (640, 480, 2000, 1002)])
def test_extract_bit_exact(ctx, oracle, w, h, nf, seed):
    img = synth.textured_frame(w, h, seed)
    p = api.default_orb_params(nf)
    kg, dg = ctx.orb_extract(img, p)
    ko, do = oracle.orb_extract(img, _oracle_params(oracle, p), cap=8192)
    assert len(ko) > nf // 4
    _same_keypoints(kg, dg, ko, do)


def test_pyramid_levels_bit_exact(ctx, oracle):
    img = synth.textured_frame(640, 480, 1001)
    p = api.default_orb_params(500)
    ctx.orb_extract(img, p)
    op = _oracle_params(oracle, p)
    for level in range(p.nlevels):
        g = ctx.orb_debug_level(level, False, 640, 480)
        o = oracle.orb_level(img, op, level, False)
        np.testing.assert_array_equal(g, o, err_msg=f"level {level}")
        gb = ctx.orb_debug_level(level, True, 640, 480)
        ob = oracle.orb_level(img, op, level, True)
        np.testing.assert_array_equal(gb, ob, err_msg=f"blurred level {level}")


def test_bgr_and_bgra_input(ctx, oracle):
    rng = np.random.default_rng(3)
    g = synth.textured_frame(400, 300, 31)
    bgr = np.stack([g, np.roll(g, 3, axis=1), np.roll(g, 5, axis=0)], axis=2)
    bgra = np.concatenate([bgr, rng.integers(0, 256, size=(300, 400, 1), dtype=np.uint8)], axis=2)
    p = api.default_orb_params(400)
    for im in (bgr, bgra):
        kg, dg = ctx.orb_extract(np.ascontiguousarray(im), p)
        ko, do = oracle.orb_extract(np.ascontiguousarray(im), _oracle_params(oracle, p), cap=8192)
        assert len(ko) > 50
        _same_keypoints(kg, dg, ko, do)


def test_fast_score_mode_and_other_params(ctx, oracle):
    img = synth.textured_frame(640, 480, 77)
    for kw in (dict(score_type=1), dict(nlevels=4, scale_factor=1.5), dict(fast_threshold=35, nfeatures=150),
               dict(edge_threshold=40), dict(nlevels=1)):
        p = api.default_orb_params(500)
        for k, v in kw.items():
            setattr(p, k, v)
        kg, dg = ctx.orb_extract(img, p)
        ko, do = oracle.orb_extract(img, _oracle_params(oracle, p), cap=8192)
        assert len(ko) > 20, kw
        _same_keypoints(kg, dg, ko, do)


def test_flat_and_tiny_frames(ctx, oracle):
    p = api.default_orb_params(500)
    flat = np.full((480, 640), 90, np.uint8)
    kg, dg = ctx.orb_extract(flat, p)
    assert len(kg) == 0
    tiny = synth.textured_frame(70, 66, 5)   # most levels are smaller than 2*edge: no keypoints there
    kg, dg = ctx.orb_extract(tiny, p)
    ko, do = oracle.orb_extract(tiny, _oracle_params(oracle, p), cap=8192)
    _same_keypoints(kg, dg, ko, do)


def test_response_ties_are_all_kept_then_overflow_is_an_error(ctx, oracle):
    """A grid of identical dots makes 936 level-0 corners tie on FAST score and Harris response:
    retainBest keeps every tie (n_out > nfeatures); if they do not fit the capacity both sides fail."""
    yy, xx = np.mgrid[0:480, 0:640]
    board = np.where(((yy % 16) == 8) & ((xx % 16) == 8), 220, 20).astype(np.uint8)
    p = api.default_orb_params(50)
    kg, dg = ctx.orb_extract(board, p)
    ko, do = oracle.orb_extract(board, _oracle_params(oracle, p), cap=8192)
    assert len(ko) == 976 and (ko["octave"] == 0).sum() == 936
    _same_keypoints(kg, dg, ko, do)
    with pytest.raises(api.MslamError) as e:
        ctx.orb_extract(board, p, cap=64)
    assert e.value.status == api.ERR_CAPACITY
    with pytest.raises(RuntimeError):
        oracle.orb_extract(board, _oracle_params(oracle, p), cap=64)


def test_more_score_ties_than_the_lds_lists_hold(oracle):
    """936 level-0 corners tie on FAST score (so the first retainBest keeps them all) but differ in Harris response (pixels at
    the corners of the 9x9 Harris window, outside the FAST ring, differ by a few grey levels): with a context of 300 keypoints
    select_k cannot keep the kept candidates in LDS and goes through its global lists; the result is the oracle's."""
    yy, xx = np.mgrid[0:480, 0:640]
    board = np.where(((yy % 16) == 8) & ((xx % 16) == 8), 220, 20).astype(np.uint8)
    rng = np.random.default_rng(12)
    for cy in range(8, 480, 16):
        for cx in range(8, 640, 16):
            for dy, dx in ((-4, -4), (-4, 4), (4, -4), (4, 4)):
                board[cy + dy, cx + dx] = 20 + int(rng.integers(0, 8))
    small = api.Context(device=0, max_w=640, max_h=480, max_kpts=300, batch=1)
    try:
        p = api.default_orb_params(50)
        kg, dg = small.orb_extract(board, p)
        ko, do = oracle.orb_extract(board, _oracle_params(oracle, p), cap=8192)
        assert 0 < (ko["octave"] == 0).sum() < 100 and len(ko) <= 300
        _same_keypoints(kg, dg, ko, do)
    finally:
        small.close()


def test_noise_frame_with_far_more_candidates_than_select_keeps_in_registers(ctx, oracle):
    """A 1080p frame of independent random pixels: several hundred thousand FAST maxima on level 0, far beyond the 16384
    candidates select_k reads into registers, so its remainder loops over the global list run; result = the oracle's."""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (1080, 1920), dtype=np.uint8)
    p = api.default_orb_params(2000)
    kg, dg = ctx.orb_extract(img, p)
    ko, do = oracle.orb_extract(img, _oracle_params(oracle, p), cap=8192)
    assert len(ko) >= 2000
    _same_keypoints(kg, dg, ko, do)


def test_tiles_where_more_than_half_the_pixels_survive_the_high_speed_test(ctx, oracle):
    """fast_nms_k appends the corners of a tile to the upper half of its survivor list; a tile with more survivors than half the
    list holds walks the survivor list again instead.  A one-pixel checkerboard passes the four-point test at every pixel (the
    pixels 3 rows / columns away have the opposite colour) without being a corner; blobs and rectangles on top give real corners."""
    img = synth.textured_frame(960, 540, 31).copy()
    yy, xx = np.mgrid[0:540, 0:960]
    board = (((xx + yy) & 1) * 200 + 20).astype(np.uint8)
    img[100:400, 150:700] = board[100:400, 150:700]          # several whole 128 x 64 tiles of level 0 and of the next levels
    img[180:260, 300:420] = synth.textured_frame(960, 540, 32)[180:260, 300:420]   # an island of texture inside the board
    p = api.default_orb_params(1500)
    kg, dg = ctx.orb_extract(img, p)
    ko, do = oracle.orb_extract(img, _oracle_params(oracle, p), cap=8192)
    assert len(ko) >= 1000
    _same_keypoints(kg, dg, ko, do)


def test_batched_device_extract(ctx, oracle):
    B, w, h, cap = 4, 640, 480, 1024
    frames = np.stack([synth.textured_frame(w, h, 100 + i) for i in range(B)])
    p = api.default_orb_params(500)
    dimg = api.DevBuf.from_array(ctx, frames)
    dk, dd = api.DevBuf(ctx, B * cap * 24), api.DevBuf(ctx, B * cap * 32)
    dn, ds = api.DevBuf(ctx, B * 4), api.DevBuf(ctx, B * 4)
    ctx.orb_extract_batch_dev(dimg.ptr, B, w, h, w, w * h, 1, p, dk.ptr, dd.ptr, cap, dn.ptr, ds.ptr)
    ctx.sync()
    n = dn.download(np.int32, (B,))
    assert (ds.download(np.int32, (B,)) == 0).all()
    kall = dk.download(api.KP_DTYPE, (B, cap))
    dall = dd.download(np.uint8, (B, cap, 32))
    for b in range(B):
        ko, do = oracle.orb_extract(frames[b], _oracle_params(oracle, p), cap=8192)
        _same_keypoints(kall[b, : n[b]], dall[b, : n[b]], ko, do)


def test_4k_frame_and_wide_scale_factors(oracle):
    """Largest frame class (3840x2160, own context) and pyramid scale factors whose tile footprint exceeds the LDS form of
    resize_k (generic path), plus a mostly flat frame with a few textured islands (sparse FAST survivor lists)."""
    with api.Context(device=0, max_w=3840, max_h=2160, max_kpts=8192, batch=1) as c4:
        img = synth.textured_frame(3840, 2160, 4242)
        p = api.default_orb_params(4000)
        kg, dg = c4.orb_extract(img, p)
        ko, do = oracle.orb_extract(img, _oracle_params(oracle, p), cap=16384)
        assert len(ko) > 3000
        _same_keypoints(kg, dg, ko, do)
        small = synth.textured_frame(1280, 720, 99)
        for sf, nl in ((1.7, 5), (2.0, 4), (1.05, 3)):
            q = api.default_orb_params(800)
            q.scale_factor, q.nlevels = sf, nl
            kg, dg = c4.orb_extract(small, q)
            ko, do = oracle.orb_extract(small, _oracle_params(oracle, q), cap=8192)
            assert len(ko) > 200, sf
            _same_keypoints(kg, dg, ko, do)
        flat = np.full((1080, 1920), 90, np.uint8)
        rng = np.random.default_rng(8)
        for _ in range(12):
            y, x = int(rng.integers(60, 900)), int(rng.integers(60, 1700))
            flat[y:y + 96, x:x + 128] = synth.textured_frame(128, 96, int(rng.integers(1, 1000)))
        kg, dg = c4.orb_extract(flat, api.default_orb_params(1000))
        ko, do = oracle.orb_extract(flat, _oracle_params(oracle, api.default_orb_params(1000)), cap=8192)
        assert 100 < len(ko)
        _same_keypoints(kg, dg, ko, do)


def test_graph_replay_of_the_extract_chain(oracle):
    """mslam_set_graphs: the captured chain replays with identical results, for several argument sets (double-buffered
    outputs, new frame contents in the same buffers) and after the plan is rebuilt for another geometry."""
    with api.Context(device=0, max_w=1280, max_h=720, max_kpts=2048, batch=2) as c:
        c.set_graphs(True)
        B, w, h, cap = 2, 640, 480, 1024
        p = api.default_orb_params(500)
        dimg = api.DevBuf(c, B * w * h)
        outs = [(api.DevBuf(c, B * cap * 24), api.DevBuf(c, B * cap * 32), api.DevBuf(c, B * 4), api.DevBuf(c, B * 4)) for _ in range(2)]
        for step in range(6):                      # steps 0/1 capture (two output sets), 2.. replay
            frames = np.stack([synth.textured_frame(w, h, 700 + 10 * step + i) for i in range(B)])
            dimg.upload(frames)
            dk, dd, dn, ds = outs[step % 2]
            c.orb_extract_batch_dev(dimg.ptr, B, w, h, w, w * h, 1, p, dk.ptr, dd.ptr, cap, dn.ptr, ds.ptr)
            c.sync()
            n = dn.download(np.int32, (B,))
            kall, dall = dk.download(api.KP_DTYPE, (B, cap)), dd.download(np.uint8, (B, cap, 32))
            for b in range(B):
                ko, do = oracle.orb_extract(frames[b], _oracle_params(oracle, p), cap=8192)
                _same_keypoints(kall[b, : n[b]], dall[b, : n[b]], ko, do)
        # another geometry: the plan is rebuilt and the cached graphs are dropped
        img = synth.textured_frame(800, 600, 3)
        for _ in range(3):
            kg, dg = c.orb_extract(img, api.default_orb_params(300))
            ko, do = oracle.orb_extract(img, _oracle_params(oracle, api.default_orb_params(300)), cap=8192)
            _same_keypoints(kg, dg, ko, do)
        c.set_graphs(False)
        kg, dg = c.orb_extract(img, api.default_orb_params(300))
        _same_keypoints(kg, dg, ko, do)


def test_unsupported_parameters_and_sizes_are_errors(ctx):
    """Parameters outside the restated ORB (WTA_K != 2, patch size != 31, first level != 0, ...) and frames beyond the
    context's capacity are refused with a status, never computed approximately."""
    img = synth.textured_frame(320, 240, 3)
    for kw in (dict(wta_k=3), dict(patch_size=25), dict(first_level=1), dict(nlevels=0), dict(nlevels=17), dict(scale_factor=1.0),
               dict(edge_threshold=10), dict(fast_threshold=0), dict(score_type=2), dict(nfeatures=-1)):
        p = api.default_orb_params(300)
        for k, v in kw.items():
            setattr(p, k, v)
        with pytest.raises(api.MslamError) as e:
            ctx.orb_extract(img, p)
        assert e.value.status == api.ERR_INVALID, kw
    big = np.zeros((1081, 1920), np.uint8)
    with pytest.raises(api.MslamError) as e:
        ctx.orb_extract(big, api.default_orb_params(300))
    assert e.value.status == api.ERR_CAPACITY
    with pytest.raises(api.MslamError) as e:                       # too small for the 16 x 16 minimum
        ctx.orb_extract(np.zeros((12, 12), np.uint8), api.default_orb_params(300))
    assert e.value.status == api.ERR_INVALID
    kg, dg = ctx.orb_extract(img, api.default_orb_params(300))     # the context is still usable afterwards
    assert len(kg) > 50
