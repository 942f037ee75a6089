"""The C-ABI collective (mslam_comm_*, mslam_allgather_i32) on RCCL, world size 1 on the GPU box: the communicator is
created from a unique id, the int32 all-gather runs on the context's stream behind the scoring kernels (device-pointer
form) and through host pointers, and the single-process form works for one device.  The merge logic over > 1 ranks is
covered on CPU by tests/test_abi_and_dist_cpu.py (gloo, world size 2)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_comm_allgather_world1_on_context_stream(ctx, oracle):
    from monocular_slam_amd import api, synth
    uid = api.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = api.Comm(ctx, uid, 0, 1)
    try:
        info = comm.info()
        assert info["rank"] == 0 and info["nranks"] == 1 and info["rccl_version"] > 20000
        send = np.arange(4096, dtype=np.int32) * 3 - 7
        assert (comm.allgather_i32(send) == send).all()
        # device-pointer form, ordered behind a PnP scoring launch on the same stream without a host sync in between
        pr = synth.pnp_problem(500, 3)
        S = synth.sample_table(500, 512, 4, 5)
        pp = api.default_pnp_params(); pp.reproj_thr = 2.0
        D = api.DevBuf.from_array
        X, x, K, dist, Sd = D(ctx, pr["X"]), D(ctx, pr["x"]), D(ctx, pr["K"]), D(ctx, pr["dist"]), D(ctx, S)
        Pn = D(ctx, np.array([500], np.int32))
        Rt, inl, counts = api.DevBuf(ctx, 96), api.DevBuf(ctx, 500), api.DevBuf(ctx, 512 * 4)
        best, ninl, gathered = api.DevBuf(ctx, 4), api.DevBuf(ctx, 4), api.DevBuf(ctx, 512 * 4)
        ctx.pnp_ransac_batch_dev(X.ptr, x.ptr, Pn.ptr, 500, K.ptr, dist.ptr, Sd.ptr, 512, 1, pp, Rt.ptr, 0, 0, inl.ptr, counts.ptr,
                                 best.ptr, ninl.ptr)
        comm.allgather_i32_dev(counts.ptr, 512, gathered.ptr)
        ctx.sync()
        opp = oracle.pnp_default_params(); opp.reproj_thr = 2.0
        o = oracle.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, opp)
        g = gathered.download(np.int32, (512,))
        assert (g == o["counts"]).all()
        b, c = api.C.c_int(), api.C.c_int()
        assert api.lib().mslam_argmax_counts(g.ctypes.data_as(api.i32p), 512, api.C.byref(b), api.C.byref(c)) == 0
        assert b.value == o["best_h"]
    finally:
        comm.close()


def test_single_process_allgather_one_device(ctx):
    from monocular_slam_amd import api
    send = np.arange(33, dtype=np.int32).reshape(1, 33) + 5
    assert (api.allgather_i32_single_process([ctx], send) == send.reshape(-1)).all()
    # invalid: the same device twice
    with pytest.raises(api.MslamError):
        api.allgather_i32_single_process([ctx, ctx], np.zeros((2, 4), np.int32))


def test_sharded_pnp_over_the_native_communicator(ctx, oracle):
    """distributed.sharded_pnp_ransac with an api.Comm (world size 1 here): slice scoring, RCCL all-gather of the padded count
    slices, global winner and its refinement equal the unsharded call."""
    from monocular_slam_amd import api, synth, distributed as D
    pr = synth.pnp_problem(400, 17)
    S = synth.sample_table(400, 301, 4, 3)
    pp = api.default_pnp_params(); pp.reproj_thr = 2.0
    comm = api.Comm(ctx, api.Comm.unique_id(), 0, 1)
    try:
        lo, hi, local, full = D.sharded_pnp_ransac(ctx, pr["X"], pr["x"], pr["K"], pr["dist"], S, pp, comm=comm)
        assert (lo, hi) == (0, 301) and (full == local).all()
        out = D.finish_sharded_pnp(ctx, pr["X"], pr["x"], pr["K"], pr["dist"], S, pp, full)
        ref = ctx.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, pp)
        assert out["best_h"] == ref["best_h"] and (full == ref["counts"]).all()
        np.testing.assert_array_equal(out["Rt"], ref["Rt"])
    finally:
        comm.close()
