"""Helper of test_gpu_parity.py::test_group_push_back_device_orders_after_producer_stream
(run as a script: `python tests/stream_order_case.py side|default`).

The images are PRODUCED on a torch stream -- a long spin, then the copy into the buffer the
group reads -- and the group is only told the stream (vh_group_set_stream); for `default` that
is handle 0, the legacy default stream.  Without the ordering the detection would read the
zero-filled buffer.  vh_group_stream_wait_images lets the producer overwrite the buffer for the
next frame without a host sync.  Results must equal the oracle's, bit for bit."""
import os
import sys

import numpy as np
import torch  # before the product library: both then share torch's HIP runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main(producer: str) -> None:
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pkg, ob = entry.load_package(), entry.load_oracle()
    oracle = ob.Oracle()
    S, W, H, T = 3, 320, 160, 4
    bpl = pkg.synth.bytes_per_line(W)
    dims = [W, H, bpl]
    seqs = [pkg.synth.stereo_sequence(W, H, T, disparity=4 + s, blur=4, seed=120 + s) for s in range(S)]
    host = np.zeros((T, 2, S, H, bpl), np.uint8)
    for s in range(S):
        for t in range(T):
            host[t, 0, s], host[t, 1, s] = seqs[s][t]
    frames = torch.from_numpy(host).to(dev)
    po = ob.Params.default()
    side = torch.cuda.Stream(device=dev) if producer == "side" else torch.cuda.default_stream(dev)
    handle = side.cuda_stream
    assert (handle == 0) == (producer == "default"), handle
    live = torch.zeros((2, S, H, bpl), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    g = pkg.StreamGroup(S, pkg.Params.default())
    g.setStream(handle)
    for t in range(T):
        with torch.cuda.stream(side):
            g.streamWaitImages(handle)     # the previous frame's detection has consumed `live`
            live.zero_()
            torch.cuda._sleep(20_000_000)  # ~10 ms: the copy below lags far behind the host
            live.copy_(frames[t])
        g.pushBackDevice(live[0].data_ptr(), live[1].data_ptr(), H * bpl, dims, False)
        if t:
            g.matchFeatures(pkg.METHOD_QUAD)
    for s in range(S):
        f = [oracle.compute_features(po, host[t_, c, s], dims)[1] for t_ in (T - 2, T - 1) for c in (0, 1)]
        for k in range(4):
            assert np.array_equal(g.getFeatures(s, k), f[k]), (producer, s, k)
        want = oracle.matching(po, dims, 2, *f)
        assert len(want) > 50 and g.getMatches(s).tobytes() == want.tobytes(), (producer, s)
    g.setStream(None)  # vh_group_clear_stream
    g.close()
    print("stream-order ok", producer)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "side")
