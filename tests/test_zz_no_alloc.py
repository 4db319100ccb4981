"""-m gpu: bt_dense_backward_static (the entry point that allocates nothing: its partial sums live in a file-scope
device array inside the library) against bt_dense_backward with a caller workspace. Bar: bit-identical -- the
same kernels in the same order, only the scratch address differs. Two calls in a row, the second with a smaller input,
so stale contents of the shared array from the first call would show."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from bullet_amd import lib  # noqa: E402
import gpu_util as gu  # noqa: E402

L = lib.load()


def _run(d, w, x, bk, y, gy, static):
    rows = d.out * d.buckets
    tw, tx, tbk, ty, tgy = gu.dev(w), gu.dev(x), gu.dev(bk), gu.dev(y), gu.dev(gy)
    tdw = torch.full((rows * d.in_,), float("nan"), device="cuda")
    tdb = torch.full((rows,), float("nan"), device="cuda")
    tdx = torch.full((d.batch, d.in_), float("nan"), device="cuda")
    if static:
        lib.check(L.bt_dense_backward_static(gu.stream(), C.byref(d), tw.data_ptr(), tx.data_ptr(), tbk.data_ptr(),
                                             ty.data_ptr(), tgy.data_ptr(), tdw.data_ptr(), tdb.data_ptr(),
                                             tdx.data_ptr()))
    else:
        wsb = L.bt_dense_backward_workspace_bytes(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        lib.check(L.bt_dense_backward(gu.stream(), C.byref(d), tw.data_ptr(), tx.data_ptr(), tbk.data_ptr(),
                                      ty.data_ptr(), tgy.data_ptr(), tdw.data_ptr(), tdb.data_ptr(), tdx.data_ptr(),
                                      ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    return tdw.cpu().numpy(), tdb.cpu().numpy(), tdx.cpu().numpy()


def _case(rng, batch, in_, out, buckets, act):
    rows = out * buckets
    d = lib.DenseDesc(batch=batch, in_=in_, out=out, buckets=buckets, act=lib.ACT[act])
    w = (rng.standard_normal(rows * in_) / np.sqrt(in_)).astype(np.float32)
    x = rng.random((batch, in_)).astype(np.float32)
    bk = rng.integers(0, buckets, size=batch).astype(np.int32)
    y = rng.random((batch, out)).astype(np.float32)
    gy = rng.standard_normal((batch, out)).astype(np.float32)
    return d, w, x, bk, y, gy


# (first call, second smaller call) per kernel family of bt_dense_backward: head (out = 1), mid-size bucketed, generic
PAIRS = [((3000, 2048, 1, 8, "identity"), (700, 512, 1, 8, "identity")),
         ((1111, 256, 5, 3, "crelu"), (130, 256, 5, 3, "crelu")),
         ((900, 20, 3, 2, "crelu"), (33, 20, 3, 2, "crelu"))]


@pytest.mark.parametrize("first,second", PAIRS)
def test_static_scratch_matches_workspace_variant(first, second):
    rng = np.random.default_rng(first[0] + first[1])
    for shape in (first, second):  # two calls in a row on the shared array, the second smaller
        case = _case(rng, *shape)
        ref = _run(*case, static=False)
        got = _run(*case, static=True)
        for a, b, what in zip(got, ref, ("dW", "dbias", "dX")):
            assert not np.isnan(a).any(), what
            assert np.array_equal(a, b), f"{what} differs for {shape}"


def test_static_scratch_rejects_oversized_shapes():
    d = lib.DenseDesc(batch=1 << 20, in_=2048, out=1, buckets=8, act=0)
    assert L.bt_dense_backward_workspace_bytes(C.byref(d)) > 16 << 20
    x = torch.zeros(16, device="cuda")
    p = x.data_ptr()
    rc = L.bt_dense_backward_static(gu.stream(), C.byref(d), p, p, p, p, p, p, p, None)
    assert rc == 1 and b"static buffer holds" in L.bt_last_error()
