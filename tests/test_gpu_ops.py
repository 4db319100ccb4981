"""-m gpu parity tests: every operator of the C ABI against the CPU oracle on the same seeded inputs.
Tolerances: integer / index work is bit-exact; fp32 sums whose order differs from the oracle's
(reference device order is itself unspecified: atomics / hipBLAS) are compared with
rtol 1e-5 relative to the result plus atol 2e-6 x (sum of |terms|), stated per test."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import oracle_binding as ob  # noqa: E402
from bullet_amd import lib  # noqa: E402
import gpu_util as gu  # noqa: E402

L = lib.load()


def test_device_is_gfx950_and_runtime_roundtrip():
    name, arch = C.create_string_buffer(128), C.create_string_buffer(128)
    wave, cus = C.c_int(), C.c_int()
    lib.check(L.bt_device_props(0, name, 128, arch, 128, C.byref(wave), C.byref(cus)))
    assert arch.value.startswith(b"gfx950") and wave.value == 64
    # malloc zeroes (Buffer::zeroed), h2d/d2h round trip (crates/gpu/src/runtime.rs:456-494 style)
    p, s = C.c_void_p(), C.c_void_p()
    lib.check(L.bt_malloc(C.byref(p), 4096))
    lib.check(L.bt_stream_create(C.byref(s)))
    host = np.ones(1024, np.float32)
    lib.check(L.bt_memcpy_d2h_async(host.ctypes.data, p, 4096, s))
    lib.check(L.bt_stream_sync(s))
    assert not host.any()
    src = np.arange(1024, dtype=np.float32)
    lib.check(L.bt_memcpy_h2d_async(p, src.ctypes.data, 4096, s))
    lib.check(L.bt_memcpy_d2h_async(host.ctypes.data, p, 4096, s))
    lib.check(L.bt_stream_sync(s))
    assert np.array_equal(host, src)
    lib.check(L.bt_stream_destroy(s))
    lib.check(L.bt_free(p))


def test_ft_forward_reference_golden_vector():
    # sparse.rs:286-298 through the HIP kernel (scalar path: rows = 2)
    d = lib.FtDesc(batch=2, rows=2, cols=3, stride=2, offset=0, nnz=4, act=0, pairwise=0, n_inputs=1, out_ld=2)
    out, _ = gu.ft_forward(d, np.arange(6, dtype=np.float32), None, [np.array([0, 1, -1, -1, 2, 2, 1, 0], np.int32)])
    assert out.reshape(-1).tolist() == [2.0, 4.0, 10.0, 14.0]


def test_ft_backward_reference_golden_vector():
    # sparse.rs:300-313 (duplicate index 2,2 in one position included)
    d = lib.FtDesc(batch=2, rows=3, cols=3, stride=3, offset=0, nnz=4, act=0, pairwise=0, n_inputs=1, out_ld=3)
    g = np.arange(6, dtype=np.float32).reshape(2, 3)
    dw, db = gu.ft_backward(d, [np.array([0, 1, -1, -1, 2, 2, 1, 0], np.int32)], g, np.ones((2, 3), np.float32), None)
    assert dw.tolist() == [3, 5, 7, 3, 5, 7, 6, 8, 10]
    assert db.tolist() == [3, 5, 7]


FT_CASES = [
    # batch rows cols nnz act pairwise n_inputs stride offset
    (300, 32, 768, 32, "screlu", 0, 2, 32, 0),       # test1 shape, ragged batch (2 chunks, tail)
    (1000, 128, 768, 32, "crelu", 0, 2, 128, 0),     # examples/simple.rs
    (513, 1024, 768, 32, "screlu", 0, 2, 1024, 0),   # config 2 FT
    (257, 256, 147, 49, "screlu", 0, 2, 256, 0),     # ataxx: nnz 49 (not a power of two)
    (400, 64, 12288, 32, "crelu", 1, 2, 64, 0),      # HalfKA-style F, pairwise
    (129, 512, 8448, 64, "crelu", 1, 2, 512, 0),     # factorised: nnz 64
    (70, 16, 100, 8, "relu", 0, 1, 48, 16),          # sliced weights: stride/offset (foldrules.rs:255-262)
    (64, 6, 50, 5, "sqrrelu", 0, 1, 6, 0),           # scalar path (rows % 4 != 0)
    (1, 8, 10, 4, "identity", 0, 1, 8, 0),           # single position
]


@pytest.fixture(params=["auto", "generic"])
def ft_path(request, ft_switches):
    """auto = LDS-resident kernels where the shape allows; generic = the any-shape kernels on every shape"""
    ft_switches(BT_FT_FORCE_GENERIC="1" if request.param == "generic" else "0")
    return request.param


FT_CASES += [
    (1100, 64, 768, 32, "crelu", 1, 2, 64, 0),       # LDS path, pairwise (16+16 row tiles), >4 chunks, tail
    (2049, 32, 300, 20, "screlu", 0, 1, 96, 32),     # LDS path with stride/offset, single input, nnz 20
    (600, 256, 147, 49, "relu", 0, 2, 256, 0),       # LDS path, nnz 49 -> 4 index registers
]


@pytest.mark.parametrize("batch,rows,cols,nnz,act,pairwise,n_inputs,stride,offset", FT_CASES)
def test_ft_forward_backward_vs_oracle(ft_path, batch, rows, cols, nnz, act, pairwise, n_inputs, stride, offset):
    rng = np.random.default_rng(batch * 7 + rows)
    out_w = rows // 2 if pairwise else rows
    d = lib.FtDesc(batch=batch, rows=rows, cols=cols, stride=stride, offset=offset, nnz=nnz, act=lib.ACT[act],
                   pairwise=pairwise, n_inputs=n_inputs, out_ld=n_inputs * out_w)
    w = (rng.standard_normal(stride * cols) * 0.2).astype(np.float32)
    bias = (rng.standard_normal(rows) * 0.3).astype(np.float32)
    idxs = [gu.random_indices(rng, batch, nnz, cols, allow_dups=(k == 1 and cols < 200)) for k in range(n_inputs)]
    idxs[0][0, :] = -1  # an empty position
    if batch > 2:
        idxs[0][1, :] = rng.choice(cols, size=nnz, replace=cols < nnz)  # a full position (no terminator)
        idxs[0][2, 0] = cols  # out-of-range (>= cols) also terminates, write.rs:231
    out, saved = gu.ft_forward(d, w, bias, idxs)
    ref_out, ref_saved = gu.oracle_ft_forward(d, w, bias, idxs)
    # same per-element summation order as the oracle (index order from 0.0f, then bias, then act) => tight
    gu.assert_close(out, ref_out, rtol=2e-6, atol=1e-6, what="ft out")
    if pairwise:
        gu.assert_close(saved, ref_saved, rtol=2e-6, atol=1e-6, what="ft saved")

    grad_out = rng.standard_normal((batch, d.out_ld)).astype(np.float32)
    dw, db = gu.ft_backward(d, idxs, grad_out, ref_out, ref_saved)
    ref_dw, ref_db, gs = gu.oracle_ft_backward(d, idxs, grad_out, ref_out, ref_saved)
    # order of the batch sum differs from the oracle's: tolerance scaled by the magnitude of the terms
    scale = float(np.abs(np.concatenate(gs)).max()) * np.sqrt(batch * n_inputs)
    gu.assert_close(dw, ref_dw, rtol=1e-5, atol=2e-6 * scale, what="dW")
    gu.assert_close(db, ref_db, rtol=1e-5, atol=2e-6 * scale, what="dbias")
    # determinism: bit-identical on a second run
    dw2, db2 = gu.ft_backward(d, idxs, grad_out, ref_out, ref_saved)
    assert np.array_equal(dw, dw2) and np.array_equal(db, db2)
    # linearity property (size-independent): backward(2g) == 2*backward(g) exactly in fp32
    dw3, _ = gu.ft_backward(d, idxs, (2 * grad_out).astype(np.float32), ref_out, ref_saved)
    assert np.array_equal(dw3, 2 * dw)


SEG_CASES = [
    # batch rows seg nseg nnz act pairwise n_inputs stride offset skew
    (700, 64, 768, 4, 32, "crelu", 1, 2, 64, 0, False),      # pairwise, 4 king buckets
    (1500, 128, 768, 16, 32, "screlu", 0, 2, 128, 0, True),  # HalfKA-16 shape, skewed buckets, some empty
    (2100, 32, 768, 10, 32, "screlu", 0, 1, 96, 32, True),   # single input, sliced weights (stride/offset)
    (520, 64, 64, 8, 8, "relu", 0, 2, 64, 0, False),         # small segments (whole table would fit: forced)
    # the FT of BASELINE configs[3] (HalfKA-16 x 1536 pairwise, 4_multi_layer.rs:79-81): 48 row tiles of 16 + 16 rows
    (1300, 1536, 768, 16, 32, "crelu", 1, 2, 1536, 0, False),
    (900, 1536, 768, 16, 32, "crelu", 1, 2, 1536, 0, True),  # the same with skewed king buckets
]


@pytest.mark.parametrize("batch,rows,seg,nseg,nnz,act,pairwise,n_inputs,stride,offset,skew", SEG_CASES)
def test_ft_segment_grouped_vs_oracle(ft_switches, batch, rows, seg, nseg, nnz, act, pairwise, n_inputs, stride, offset,
                                      skew):
    """bt_ft_desc.segment: tables too large for LDS processed segment by segment (entry lists + the LDS kernels).
    Same bars as the ungrouped kernels: forward keeps the per-element summation order, backward is a batch sum."""
    # several forward / backward items per segment at this batch size
    ft_switches(BT_FT_FORCE_GROUPED=1, BT_FT_FWD_ITEM=256, BT_FT_BWD_CHUNKS=2)
    rng = np.random.default_rng(batch + rows)
    cols = seg * nseg
    out_w = rows // 2 if pairwise else rows
    d = lib.FtDesc(batch=batch, rows=rows, cols=cols, stride=stride, offset=offset, nnz=nnz, act=lib.ACT[act],
                   pairwise=pairwise, n_inputs=n_inputs, out_ld=n_inputs * out_w, segment=seg)
    w = (rng.standard_normal(stride * cols) * 0.2).astype(np.float32)
    bias = (rng.standard_normal(rows) * 0.3).astype(np.float32)
    probs = None
    if skew:
        probs = np.zeros(nseg)
        probs[[0, 1, nseg - 1]] = [0.6, 0.3, 0.1]  # most segments empty, one holding 60 % of the entries
    idxs = [gu.segment_indices(rng, batch, nnz, nseg, seg, probs) for _ in range(n_inputs)]
    idxs[0][0, :] = -1  # an empty position
    idxs[0][1, :] = (nseg - 1) * seg + rng.choice(seg, size=nnz, replace=False)  # a full one (no terminator)
    idxs[0][2, 0] = cols  # out-of-range first index terminates the (empty) list
    out, saved = gu.ft_forward(d, w, bias, idxs)
    assert gu.ft_forward.last_violations == 0
    ref_out, ref_saved = gu.oracle_ft_forward(d, w, bias, idxs)
    gu.assert_close(out, ref_out, rtol=2e-6, atol=1e-6, what="ft out")
    if pairwise:
        gu.assert_close(saved, ref_saved, rtol=2e-6, atol=1e-6, what="ft saved")
    grad_out = rng.standard_normal((batch, d.out_ld)).astype(np.float32)
    dw, db = gu.ft_backward(d, idxs, grad_out, ref_out, ref_saved)
    ref_dw, ref_db, gs = gu.oracle_ft_backward(d, idxs, grad_out, ref_out, ref_saved)
    scale = float(np.abs(np.concatenate(gs)).max()) * np.sqrt(batch * n_inputs)
    gu.assert_close(dw, ref_dw, rtol=1e-5, atol=2e-6 * scale, what="dW")
    gu.assert_close(db, ref_db, rtol=1e-5, atol=2e-6 * scale, what="dbias")
    dw2, db2 = gu.ft_backward(d, idxs, grad_out, ref_out, ref_saved)
    assert np.array_equal(dw, dw2) and np.array_equal(db, db2)  # bit-identical run to run
    # the same table without the promise (any-shape kernels): same numbers within the summation-order bar
    d0 = lib.FtDesc(batch=batch, rows=rows, cols=cols, stride=stride, offset=offset, nnz=nnz, act=lib.ACT[act],
                    pairwise=pairwise, n_inputs=n_inputs, out_ld=n_inputs * out_w, segment=0)
    out0, _ = gu.ft_forward(d0, w, bias, idxs)
    assert np.array_equal(out0, out)  # forward: identical order => identical bits


def test_ft_segment_promise_violation_is_loud(ft_switches):
    """an index outside its entry's segment: counted on the device and the entry's output turns into NaN"""
    ft_switches(BT_FT_FORCE_GROUPED=1)
    rng = np.random.default_rng(5)
    d = lib.FtDesc(batch=300, rows=32, cols=4 * 768, stride=32, offset=0, nnz=32, act=lib.ACT["crelu"], pairwise=0,
                   n_inputs=1, out_ld=32, segment=768)
    w = rng.standard_normal(32 * d.cols).astype(np.float32)
    idx = gu.segment_indices(rng, 300, 32, 4, 768, min_cnt=4)
    idx[17, 2] = (idx[17, 0] // 768 * 768 + 768 + 5) % d.cols  # third feature of position 17 from another segment
    out, _ = gu.ft_forward(d, w, None, [idx])
    assert gu.ft_forward.last_violations >= 1
    assert np.isnan(out[17]).all() and not np.isnan(np.delete(out, 17, axis=0)).any()


DENSE_CASES = [(777, 2048, 1, 8, "identity"), (300, 1024, 16, 8, "screlu"), (300, 16, 32, 8, "screlu"),
               (300, 32, 1, 8, "identity"), (1000, 64, 1, 1, "identity"), (50, 512, 1, 1, "identity"),
               (33, 20, 3, 2, "crelu"), (300, 8, 3, 8, "identity"),  # fewer threads (out*in) than positions per range
               # mid-size bucketed path (one bucket's rows in LDS): ragged chunks, single bucket, 2 staging rounds of columns
               (1111, 256, 5, 3, "crelu"), (130, 2048, 9, 1, "relu"), (64, 512, 32, 8, "screlu")]


@pytest.mark.parametrize("batch,in_,out,buckets,act", DENSE_CASES)
def test_dense_forward_backward_vs_oracle(batch, in_, out, buckets, act):
    rng = np.random.default_rng(in_ * 3 + out)
    rows = out * buckets
    d = lib.DenseDesc(batch=batch, in_=in_, out=out, buckets=buckets, act=lib.ACT[act])
    w = (rng.standard_normal(rows * in_) * (1.0 / np.sqrt(in_))).astype(np.float32)
    bias = (rng.standard_normal(rows) * 0.1).astype(np.float32)
    x = rng.random((batch, in_)).astype(np.float32)
    bk = rng.integers(0, buckets, size=batch).astype(np.int32)
    gy = rng.standard_normal((batch, out)).astype(np.float32)
    # ---- oracle: Matmul (all bucket rows) + bias + Select + act; backward via SelectPad + Matmul^T
    O = ob.lib()
    z_full = np.zeros((batch, rows), np.float32)
    O.orc_matmul(ob.fp(w), ob.fp(x), ob.fp(z_full), 1, rows, in_, batch, 1, 1)
    z_full = (z_full + bias[None, :]).astype(np.float32)
    z = np.zeros((batch, out), np.float32)
    O.orc_select(ob.fp(z_full), ob.ip(bk), ob.fp(z), batch, rows, buckets)
    y_ref = np.zeros_like(z)
    O.orc_act_fwd(d.act, ob.fp(z), ob.fp(y_ref), C.c_size_t(z.size))
    gz = np.zeros_like(z)
    O.orc_act_bwd(d.act, ob.fp(y_ref), ob.fp(gy), ob.fp(gz), C.c_size_t(z.size))
    gz_full = np.zeros((batch, rows), np.float32)
    O.orc_select_pad(ob.fp(gz), ob.ip(bk), ob.fp(gz_full), batch, rows, buckets)
    dw_ref = np.zeros(rows * in_, np.float32)   # dW = dY X^T : (rows x B) @ (B x in), X^T is row-major view of x
    O.orc_matmul(ob.fp(gz_full), ob.fp(x), ob.fp(dw_ref), 1, rows, batch, in_, 1, 0)
    dx_ref = np.zeros((batch, in_), np.float32)  # dX = W^T dY
    O.orc_matmul(ob.fp(w), ob.fp(gz_full), ob.fp(dx_ref), 1, in_, rows, batch, 0, 1)
    db_ref = np.zeros(rows, np.float32)
    O.orc_reduce_sum_batch(ob.fp(gz_full), ob.fp(db_ref), batch, rows)
    # ---- device
    tw, tb, tx, tbk, tgy = gu.dev(w), gu.dev(bias), gu.dev(x), gu.dev(bk), gu.dev(gy)
    ty = torch.full((batch, out), float("nan"), device="cuda")
    lib.check(L.bt_dense_forward(gu.stream(), C.byref(d), tw.data_ptr(), tb.data_ptr(), tx.data_ptr(),
                                 tbk.data_ptr(), ty.data_ptr()))
    gu.assert_close(ty.cpu().numpy(), y_ref, rtol=1e-5, atol=2e-6 * np.sqrt(in_), what="dense y")
    wsb = L.bt_dense_backward_workspace_bytes(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    tdw = torch.full((rows * in_,), float("nan"), device="cuda")
    tdb = torch.full((rows,), float("nan"), device="cuda")
    tdx = torch.full((batch, in_), float("nan"), device="cuda")
    ty_ref = gu.dev(y_ref)
    lib.check(L.bt_dense_backward(gu.stream(), C.byref(d), tw.data_ptr(), tx.data_ptr(), tbk.data_ptr(),
                                  ty_ref.data_ptr(), tgy.data_ptr(), tdw.data_ptr(), tdb.data_ptr(), tdx.data_ptr(),
                                  ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    s = float(np.abs(gz).max()) * np.sqrt(batch)
    gu.assert_close(tdw.cpu().numpy(), dw_ref, rtol=1e-5, atol=2e-6 * s, what="dense dW")
    gu.assert_close(tdb.cpu().numpy(), db_ref, rtol=1e-5, atol=2e-6 * s, what="dense db")
    gu.assert_close(tdx.cpu().numpy(), dx_ref, rtol=1e-5, atol=2e-6 * np.sqrt(out), what="dense dX")
    # unselected bucket rows get exact zeros (SelectPad, select.rs:101-104)
    dwm = tdw.cpu().numpy().reshape(in_, rows)
    for k in range(buckets):
        if not (bk == k).any():
            assert not dwm[:, k * out:(k + 1) * out].any()


@pytest.mark.parametrize("batch,use_w", [(1, False), (1000, False), (16384, True)])
def test_sigmoid_mse_vs_oracle(batch, use_w):
    rng = np.random.default_rng(batch)
    y = (rng.standard_normal(batch) * 2).astype(np.float32)
    t = rng.random(batch).astype(np.float32)
    w = (rng.random(batch) + 0.5).astype(np.float32) if use_w else None
    s = np.array([ob.lib().orc_sigmoid(float(v)) for v in y], np.float32)
    diff = (s + (-1.0 * t)).astype(np.float32)
    g = w if use_w else np.ones(batch, np.float32)
    l_ref = float(np.sum((g * diff * diff).astype(np.float64)))
    gy_ref = ((g * diff + g * diff) * (s * (1.0 + (-1.0 * s)))).astype(np.float32)
    ty, tt = gu.dev(y), gu.dev(t)
    tw = gu.dev(w) if use_w else None
    tl, tg = torch.zeros(1, device="cuda"), torch.zeros(batch, device="cuda")
    lib.check(L.bt_sigmoid_mse(gu.stream(), batch, ty.data_ptr(), tt.data_ptr(), tw.data_ptr() if use_w else None,
                               tl.data_ptr(), tg.data_ptr()))
    assert abs(tl.item() - l_ref) <= 2e-6 * max(1.0, l_ref) * np.sqrt(batch) ** 0.5
    gu.assert_close(tg.cpu().numpy(), gy_ref, rtol=2e-5, atol=1e-7, what="grad_y")


@pytest.mark.parametrize("kind,width,power,use_w", [("sigmoid_power", 1, 2.5, False), ("sigmoid_power", 1, 2.0, True),
                                                   ("softmax_ce", 3, 0.0, False), ("softmax_ce", 3, 0.0, True),
                                                   ("softmax_ce", 7, 0.0, False), ("sigmoid_mse", 1, 0.0, True)])
def test_loss_kinds_vs_oracle(kind, width, power, use_w):
    """power_error / softmax cross-entropy. expf/logf/powf are the device's libm, so a few ulp per
    element against glibc: rtol 1e-5 on the gradient, 1e-6 * sqrt(batch) relative on the summed loss."""
    batch = 3000
    rng = np.random.default_rng(width * 10 + use_w)
    y = (rng.standard_normal((batch, width)) * 2).astype(np.float32)
    if kind == "softmax_ce":
        t = np.zeros((batch, width), np.float32)
        t[np.arange(batch), rng.integers(0, width, batch)] = 1.0
        t[::7] = rng.dirichlet(np.ones(width), size=t[::7].shape[0]).astype(np.float32)  # soft targets too
    else:
        t = rng.random((batch, width)).astype(np.float32)
        t[5] = ob.lib().orc_sigmoid(float(y[5, 0]))  # diff == 0: sgn(0) = 0 branch
    w = (rng.random(batch) + 0.5).astype(np.float32) if use_w else None
    l_ref, gy_ref = ob.loss(kind, y, t, w, power)
    ty, tt = gu.dev(y), gu.dev(t)
    tw = gu.dev(w) if use_w else None
    tl, tg = torch.zeros(1, device="cuda"), torch.zeros(batch * width, device="cuda")
    lib.check(L.bt_loss(gu.stream(), lib.LOSS[kind], batch, width, ty.data_ptr(), tt.data_ptr(),
                        tw.data_ptr() if use_w else None, power, tl.data_ptr(), tg.data_ptr()))
    assert abs(tl.item() - l_ref) <= 1e-6 * np.sqrt(batch) * max(1.0, abs(l_ref))
    gu.assert_close(tg.cpu().numpy().reshape(batch, width), gy_ref, rtol=1e-5, atol=2e-7, what="grad_y")
    assert L.bt_loss(gu.stream(), 1, batch, 3, ty.data_ptr(), tt.data_ptr(), None, 2.0, tl.data_ptr(), tg.data_ptr()) == 1
    assert L.bt_loss(gu.stream(), 9, batch, 1, ty.data_ptr(), tt.data_ptr(), None, 2.0, tl.data_ptr(), tg.data_ptr()) == 1


def test_weight_clip_and_decay_vs_oracle():
    """kernels of the WeightClipping / WeightDecay wrappers (clip.rs:18-31, decay.rs:18-40): bit-exact (one rounding
    each, no contraction possible in the clip; the decay factor is formed the same way)"""
    rng = np.random.default_rng(11)
    w = (rng.standard_normal(100003) * 2).astype(np.float32)
    dw = gu.dev(w)
    lr = torch.tensor([0.00123], device="cuda")
    lib.check(L.bt_weight_decay(gu.stream(), dw.data_ptr(), w.size, 0.01, lr.data_ptr()))
    lib.check(L.bt_weight_clip(gu.stream(), dw.data_ptr(), w.size, -1.98, 1.5))
    ob.weight_decay(w, 0.01, float(np.float32(0.00123)))
    ob.weight_clip(w, -1.98, 1.5)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dw.cpu().numpy(), w)
    assert w.max() == 1.5 and w.min() == np.float32(-1.98)


def test_clip_pass_through_and_faux_quantise_vs_oracle():
    """clip_pass_through_grad / faux_quantise forward kernels (model/operations/pointwise.rs:196-221,272-300): bit-exact
    with the oracle (single roundings, no contraction possible), in place and out of place"""
    rng = np.random.default_rng(12)
    x = (rng.standard_normal(100003) * 1.5).astype(np.float32)
    x[:4] = [0.5 / 255, -0.5 / 255, 1.5 / 255, 1e9]
    dx, dy = gu.dev(x), gu.dev(np.zeros_like(x))
    lib.check(L.bt_clip_pass_through(gu.stream(), dx.data_ptr(), dy.data_ptr(), x.size, -1.98, 1.5))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dy.cpu().numpy(), ob.clip_pass_through(x, -1.98, 1.5))
    for value, rnd in ((255.0, True), (255.0, False), (64.0, True), (0.3, False)):
        lib.check(L.bt_faux_quantise(gu.stream(), dx.data_ptr(), dy.data_ptr(), x.size, value, int(rnd)))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(dy.cpu().numpy(), ob.faux_quantise(x, value, rnd))
    lib.check(L.bt_faux_quantise(gu.stream(), dx.data_ptr(), dx.data_ptr(), x.size, 255.0, 1))  # in place
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dx.cpu().numpy(), ob.faux_quantise(x, 255.0, True))
    assert L.bt_faux_quantise(gu.stream(), dx.data_ptr(), dy.data_ptr(), x.size, 0.0, 1) != 0


def test_adamw_multi_tensor_vs_oracle():
    rng = np.random.default_rng(3)
    sizes = [786432, 1024, 16384, 8, 5, 1023]  # incl. sizes not divisible by 4 (scalar tail, adam.rs:139-158)
    hp = [(0.01, 0.9, 0.999, -1.98, 1.98)] * 4 + [(0.0, 0.8, 0.99, -0.01, 0.01), (0.01, 0.9, 0.999, -0.99, 0.99)]
    P = [(rng.standard_normal(n)).astype(np.float32) for n in sizes]
    M = [(rng.standard_normal(n) * 0.01).astype(np.float32) for n in sizes]
    V = [(rng.random(n) * 1e-4).astype(np.float32) for n in sizes]
    G = [(rng.standard_normal(n) * 50).astype(np.float32) for n in sizes]
    dP, dM, dV, dG = [[gu.dev(a) for a in X] for X in (P, M, V, G)]
    arr = (lib.AdamwTensor * len(sizes))()
    for i, n in enumerate(sizes):
        arr[i] = lib.AdamwTensor(dG[i].data_ptr(), dP[i].data_ptr(), dM[i].data_ptr(), dV[i].data_ptr(), n, *hp[i])
    adj, lr = torch.tensor([1.0 / 16384], device="cuda"), torch.tensor([0.001], device="cuda")
    for _ in range(3):
        lib.check(L.bt_adamw(gu.stream(), len(sizes), arr, adj.data_ptr(), lr.data_ptr()))
        for i in range(len(sizes)):
            ob.adamw(P[i], M[i], V[i], G[i], 1.0 / 16384, 0.001, *hp[i])
    torch.cuda.synchronize()
    for i in range(len(sizes)):
        # elementwise, same operation order; only FMA contraction may differ => a few ulp
        gu.assert_close(dP[i].cpu().numpy(), P[i], rtol=2e-6, atol=1e-7, what=f"p{i}")
        gu.assert_close(dM[i].cpu().numpy(), M[i], rtol=2e-6, atol=1e-9, what=f"m{i}")
        gu.assert_close(dV[i].cpu().numpy(), V[i], rtol=2e-6, atol=1e-12, what=f"v{i}")
    bad = lib.AdamwTensor(dG[0].data_ptr(), dP[0].data_ptr(), dM[0].data_ptr(), dV[0].data_ptr(), 4, 0, .9, .99, 1., -1.)
    assert L.bt_adamw(gu.stream(), 1, C.byref(bad), adj.data_ptr(), lr.data_ptr()) == 1
    assert b"Invalid clipping" in L.bt_last_error()  # OptimiserState::new, adam.rs:259-263


def test_shape_errors_are_reported_not_launched():
    d = lib.FtDesc(batch=4, rows=8, cols=16, stride=8, offset=0, nnz=4, act=0, pairwise=0, n_inputs=2, out_ld=8)
    x = torch.zeros(64, device="cuda")
    ptrs = (C.c_void_p * 2)(x.data_ptr(), x.data_ptr())
    rc = L.bt_ft_forward(gu.stream(), C.byref(d), x.data_ptr(), None, ptrs, x.data_ptr(), None)
    assert rc == 1 and b"Mismatched sizes" in L.bt_last_error()  # out_ld < n_inputs * rows
    d.out_ld = 16
    rc = L.bt_ft_backward(gu.stream(), C.byref(d), ptrs, x.data_ptr(), x.data_ptr(), None, x.data_ptr(), None,
                          x.data_ptr(), 16)
    assert rc == 1 and b"workspace too small" in L.bt_last_error()
