"""nmtgan_logits_sample_static - the entry point that allocates nothing (its scratch is a static device array) -
against nmtgan_logits_sample with a caller workspace: same kernels, so the sampled ids must be identical.
Two calls in a row on the same GPU, the second with a smaller input (the array keeps the first call's
contents), then a shape beyond the array's size, which must be refused with a message; then, after those warm calls,
the entry point captured into a HIP graph and replayed (its callers include code under graph capture)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op", ["fp16", "fp32"])
def test_static_scratch_entry_point_equals_workspace_entry_point(cuda, op):
    from nmt_gan_amd import _lib as L
    dt, opc = L.op_dtype_of(op)
    V, d = 32000, 512
    rng = np.random.default_rng(17)
    w = torch.from_numpy((rng.standard_normal((V, d)) * 0.1).astype(np.float32)).to(dt).to(cuda)
    for M in (3000, 700):                                          # second call: smaller input, dirty scratch
        x = torch.from_numpy(rng.standard_normal((M, d)).astype(np.float32)).to(dt).to(cuda)
        u = torch.from_numpy(rng.random(M + 9, dtype=np.float32)).to(cuda)
        u_off = torch.from_numpy(rng.permutation(M).astype(np.int64)).to(cuda)
        want = torch.full((M,), -1, dtype=torch.int32, device=cuda)
        got = torch.full((M,), -2, dtype=torch.int32, device=cuda)
        nb = L.lib.nmtgan_logits_sample_workspace_bytes(M, V)
        assert 0 < nb <= (32 << 20)
        ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
        L.check(L.lib.nmtgan_logits_sample(opc, x.data_ptr(), w.data_ptr(), M, V, d, u.data_ptr(), u_off.data_ptr(), 3,
                                           want.data_ptr(), ws.data_ptr(), nb, L.stream_ptr()), "logits_sample")
        torch.cuda.synchronize()
        print("static-scratch call: op", op, "M", M, flush=True)
        L.check(L.lib.nmtgan_logits_sample_static(opc, x.data_ptr(), w.data_ptr(), M, V, d, u.data_ptr(),
                                                  u_off.data_ptr(), 3, got.data_ptr(), L.stream_ptr()),
                "logits_sample_static")
        torch.cuda.synchronize()
        assert int(want.min()) >= 0 and int(want.max()) < V
        assert torch.equal(got, want), int((got != want).sum())
    # beyond the static array: refused on the host, nothing launched
    assert L.lib.nmtgan_logits_sample_workspace_bytes(20000, V) > (32 << 20)
    rc = L.lib.nmtgan_logits_sample_static(opc, x.data_ptr(), w.data_ptr(), 20000, V, d, u.data_ptr(), None, 0,
                                           got.data_ptr(), L.stream_ptr())
    assert rc != 0 and b"static buffer holds" in L.lib.nmtgan_last_error()


def test_static_scratch_entry_point_under_graph_capture(cuda):
    """After one warm call (which resolves the scratch address) the entry point can be captured: three replays with
    eager work and synchronisations between them give the ids of the eager call each time."""
    from nmt_gan_amd import _lib as L
    dt, opc = L.op_dtype_of("fp16")
    V, d, M = 32000, 512, 1500
    rng = np.random.default_rng(23)
    w = torch.from_numpy((rng.standard_normal((V, d)) * 0.1).astype(np.float32)).to(dt).to(cuda)
    x = torch.from_numpy(rng.standard_normal((M, d)).astype(np.float32)).to(dt).to(cuda)
    u = torch.from_numpy(rng.random(M, dtype=np.float32)).to(cuda)
    want = torch.full((M,), -1, dtype=torch.int32, device=cuda)
    got = torch.full((M,), -2, dtype=torch.int32, device=cuda)

    def call(out):
        L.check(L.lib.nmtgan_logits_sample_static(opc, x.data_ptr(), w.data_ptr(), M, V, d, u.data_ptr(), None, 0,
                                                  out.data_ptr(), L.stream_ptr()), "logits_sample_static")

    call(want)                                                     # warm, eager: the reference ids
    torch.cuda.synchronize()
    assert int(want.min()) >= 0 and int(want.max()) < V
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="relaxed"):
        call(got)
    for i in range(3):
        got.fill_(-2)
        if i == 1:                                                 # an eager call in between re-uses the shared scratch
            call(torch.full((M,), -3, dtype=torch.int32, device=cuda))
        torch.cuda.synchronize()
        print("static-scratch graph replay", i, flush=True)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want), (i, int((got != want).sum()))
