"""The CPU oracle against every in-tree golden vector of the reference that pins the hot path
(SURVEY.md section 8c). Vectors are copied as data from the reference's #[test] functions."""
import ctypes as C

import numpy as np

import oracle_binding as ob

L = ob.lib()
f32 = lambda *a: np.array(a, dtype=np.float32)  # noqa: E731
i32 = lambda *a: np.array(a, dtype=np.int32)  # noqa: E731


def test_sparse_matmul_fwd_golden():
    # crates/compiler/src/tensor/operation/linear/sparse.rs:286-298
    w, idx, out = f32(0, 1, 2, 3, 4, 5), i32(0, 1, -1, -1, 2, 2, 1, 0), np.zeros(4, np.float32)
    for brk in (0, 1):  # literal (skip) and device (break) rules agree when the tail after -1 is all -1
        L.orc_sparse_matmul_fwd(ob.fp(w), ob.ip(idx), ob.fp(out), 2, 2, 3, 2, 0, 4, brk)
        assert out.tolist() == [2.0, 4.0, 10.0, 14.0]


def test_sparse_matmul_bwd_multi_golden():
    # sparse.rs:300-313
    g, idx, out = f32(0, 1, 2, 3, 4, 5), i32(0, 1, -1, -1, 2, 2, 1, 0), np.zeros(9, np.float32)
    gp = (C.POINTER(C.c_float) * 1)(ob.fp(g))
    ipp = (C.POINTER(C.c_int32) * 1)(ob.ip(idx))
    rows, offs = (C.c_int * 1)(3), (C.c_int * 1)(0)
    L.orc_sparse_matmul_bwd_multi(1, gp, ipp, ob.fp(out), 2, rows, 3, 3, offs, 4, 1)
    assert out.tolist() == [3, 5, 7, 3, 5, 7, 6, 8, 10]


def test_break_vs_skip_rule_differs_on_stale_tail():
    # slots after the first -1 hold stale indices; the device rule must stop there
    w, idx = f32(1, 10, 100), i32(0, -1, 2, 1)
    out = np.zeros(1, np.float32)
    L.orc_sparse_matmul_fwd(ob.fp(w), ob.ip(idx), ob.fp(out), 1, 1, 3, 1, 0, 4, 1)
    assert out[0] == 1.0
    L.orc_sparse_matmul_fwd(ob.fp(w), ob.ip(idx), ob.fp(out), 1, 1, 3, 1, 0, 4, 0)
    assert out[0] == 111.0


def test_matmul_golden_all_layouts():
    # crates/compiler/src/tensor/operation/linear/matmul.rs:133-174 (I32 vectors; exact in f32)
    lhs = f32(0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5)
    rhs = f32(5, 4, 3, 2, 1, 0, 5, 4, 3, 2, 1, 0)
    out = np.zeros(8, np.float32)
    for lcm, rcm, exp in [(1, 1, [20, 32, 2, 5] * 2), (0, 0, [5, 32, 2, 20] * 2), (1, 0, [10, 19, 4, 10] * 2),
                          (0, 1, [10, 46, 1, 10] * 2)]:
        L.orc_matmul(ob.fp(lhs), ob.fp(rhs), ob.fp(out), 2, 2, 3, 2, lcm, rcm)
        assert out.tolist() == exp


def test_select_golden():
    # crates/compiler/src/tensor/operation/index/select.rs:172-222
    inp, ind, out = f32(*range(8)), i32(0, 1), np.zeros(4, np.float32)
    L.orc_select(ob.fp(inp), ob.ip(ind), ob.fp(out), 2, 4, 2)
    assert out.tolist() == [0, 1, 6, 7]
    inp, out = f32(0, 1, 2, 3), np.ones(8, np.float32)
    L.orc_select_pad(ob.fp(inp), ob.ip(ind), ob.fp(out), 2, 4, 2)
    assert out.tolist() == [0, 1, 0, 0, 0, 0, 2, 3]
    inp, ind = f32(*range(64)), i32(*(list(range(4)) * 4))
    pad, inv = np.zeros(256, np.float32), np.zeros(64, np.float32)
    L.orc_select_pad(ob.fp(inp), ob.ip(ind), ob.fp(pad), 16, 16, 4)
    L.orc_select(ob.fp(pad), ob.ip(ind), ob.fp(inv), 16, 16, 4)
    assert np.array_equal(inp, inv)


def test_reduce_sum_batch_golden():
    # crates/compiler/src/tensor/operation/linear/reduce.rs:201-244, outer-dimension sum of an 8x(2*4) iota
    inp = np.arange(64, dtype=np.float32)
    out = np.zeros(8, np.float32)
    L.orc_reduce_sum_batch(ob.fp(inp), ob.fp(out), 8, 8)
    assert out.tolist() == inp.reshape(8, 8).sum(0).tolist()


def test_activation_derivatives_from_output():
    # dfo.rs:47-103: CReLU' is zero at both clip points, SCReLU' = 2*sqrt(out) inside
    x = f32(-1.0, 0.0, 0.25, 0.5, 1.0, 2.0)
    for act, fwd, der in [
        ("relu", [0, 0, .25, .5, 1, 2], [0, 0, 1, 1, 1, 1]),
        ("crelu", [0, 0, .25, .5, 1, 1], [0, 0, 1, 1, 0, 0]),
        ("screlu", [0, 0, .0625, .25, 1, 1], [0, 0, .5, 1, 0, 0]),
        ("sqrrelu", [0, 0, .0625, .25, 1, 4], [0, 0, .5, 1, 2, 4]),
    ]:
        y, g = np.zeros_like(x), np.zeros_like(x)
        L.orc_act_fwd(ob.ACT[act], ob.fp(x), ob.fp(y), C.c_size_t(x.size))
        assert y.tolist() == fwd
        L.orc_act_bwd(ob.ACT[act], ob.fp(y), ob.fp(np.ones_like(x)), ob.fp(g), C.c_size_t(x.size))
        assert g.tolist() == der


def test_adamw_order_of_operations():
    # adam.rs:43-51: decay -> moments -> step -> clamp, eps added after sqrt, no bias correction
    p, m, v, g = f32(1.0, 1.975), f32(0, 0), f32(0, 0), f32(2.0, -4.0)
    ob.adamw(p, m, v, g, adj=0.5, lr=0.1)
    grad = np.float32(0.5) * g
    em = np.float32(0.1) * grad  # (1 - beta1) * grad in f32
    ev = np.float32(1.0 - np.float32(0.999)) * grad * grad
    assert np.allclose(m, em, rtol=1e-6) and np.allclose(v, ev, rtol=1e-5)
    e0 = np.float32(1.0) * np.float32(1 - 0.01 * 0.1) - np.float32(0.1) * (em[0] / (np.sqrt(ev[0]) + np.float32(1e-8)))
    assert abs(p[0] - e0) < 1e-6
    assert p[1] == np.float32(1.98)  # clamped to max_weight


def test_material_count_bucket():
    # crates/bullet_lib/src/game/outputs.rs:24-31: (popcount - 2) / ceil(32/N)
    for n_pieces in range(2, 33):
        occ = (1 << n_pieces) - 1
        assert L.orc_material_bucket(occ, 8) == (n_pieces - 2) // 4
        assert L.orc_material_bucket(occ, 1) == 0


# ---- generic index / reduce ops: every per-dimension vector of the reference's #[test]s --------------------------
def _dims(shape, dim, include_dim_in_inner=False):
    outer = int(np.prod(shape[:dim], dtype=np.int64))
    inner = int(np.prod(shape[dim + (0 if include_dim_in_inner else 1):], dtype=np.int64))
    return outer, shape[dim], inner


_u32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
_sz = C.c_size_t


def test_broadcast_golden_and_inverse():
    # index/broadcast.rs:160-209: shapes [2,4] and [2,2,2], repeats 2; the inverse (reduce-sum) gives 2 * input
    cases = [([2, 4], 0, [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7]),
             ([2, 4], 1, [0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 6, 7, 4, 5, 6, 7]),
             ([2, 2, 2], 0, [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7]),
             ([2, 2, 2], 1, [0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 6, 7, 4, 5, 6, 7]),
             ([2, 2, 2], 2, [0, 1, 0, 1, 2, 3, 2, 3, 4, 5, 4, 5, 6, 7, 6, 7])]
    inp = np.arange(8, dtype=np.int32)
    for shape, dim, exp in cases:
        outer, _, inner = _dims(shape, dim, include_dim_in_inner=True)
        out = np.zeros(16, np.int32)
        L.orc_broadcast(_u32p(inp), _sz(outer), _sz(inner), _sz(2), _u32p(out))
        assert out.tolist() == exp, (shape, dim)
        back = np.zeros(8, np.int32)  # Broadcast <-> Reduce (broadcast.rs:121-124): sum over the repeats
        L.orc_reduce_i32(ob.ip(out), _sz(outer), _sz(2), _sz(inner), 0, ob.ip(back))
        assert back.tolist() == (2 * inp).tolist()


def test_pad_and_slice_golden():
    # index/pad.rs:178-223 (pad 1 before and after with 0) and its inverse; index/slice.rs:160-198 (slice 0..1)
    inp = np.arange(8, dtype=np.int32)
    pads = [([2, 4], 0, [0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 0, 0, 0, 0]),
            ([2, 4], 1, [0, 0, 1, 2, 3, 0, 0, 4, 5, 6, 7, 0]),
            ([2, 2, 2], 0, [0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 0, 0, 0, 0]),
            ([2, 2, 2], 1, [0, 0, 0, 1, 2, 3, 0, 0, 0, 0, 4, 5, 6, 7, 0, 0]),
            ([2, 2, 2], 2, [0, 0, 1, 0, 0, 2, 3, 0, 0, 4, 5, 0, 0, 6, 7, 0])]
    for shape, dim, exp in pads:
        outer, dimen, inner = _dims(shape, dim)
        out = np.full(len(exp), -1, np.int32)
        L.orc_pad(_u32p(inp), _sz(outer), _sz(dimen), _sz(inner), _sz(1), _sz(1), C.c_uint32(0), _u32p(out))
        assert out.tolist() == exp, (shape, dim)
        back = np.zeros(8, np.int32)  # Pad <-> Slice
        L.orc_slice(_u32p(out), _sz(outer), _sz(dimen + 2), _sz(inner), _sz(1), _sz(1 + dimen), _u32p(back))
        assert back.tolist() == inp.tolist()
    slices = [([2, 4], 0, [0, 1, 2, 3]), ([2, 4], 1, [0, 4]), ([2, 2, 2], 0, [0, 1, 2, 3]), ([2, 2, 2], 1, [0, 1, 4, 5]),
              ([2, 2, 2], 2, [0, 2, 4, 6])]
    for shape, dim, exp in slices:
        outer, dimen, inner = _dims(shape, dim)
        out = np.zeros(len(exp), np.int32)
        L.orc_slice(_u32p(inp), _sz(outer), _sz(dimen), _sz(inner), _sz(0), _sz(1), _u32p(out))
        assert out.tolist() == exp, (shape, dim)


def test_reduce_golden_all_dimensions():
    # linear/reduce.rs:201-244: shape [8, 2, 4] of iota(64), Sum / Max / Min over each dimension
    inp = np.arange(64, dtype=np.int32)
    shape = [8, 2, 4]
    expected = {
        (0, 0): lambda i: 224 + 8 * i, (1, 0): lambda i: 4 + 2 * (i % 4) + 16 * (i // 4), (2, 0): lambda i: 6 + 16 * i,
        (0, 1): lambda i: 56 + i, (1, 1): lambda i: i + 4 * (1 + i // 4), (2, 1): lambda i: 3 + 4 * i,
        (0, 2): lambda i: i, (1, 2): lambda i: i + 4 * (i // 4), (2, 2): lambda i: 4 * i,
    }
    for (dim, kind), fn in expected.items():
        outer, dimen, inner = _dims(shape, dim)
        out = np.zeros(64 // shape[dim], np.int32)
        L.orc_reduce_i32(ob.ip(inp), _sz(outer), _sz(dimen), _sz(inner), kind, ob.ip(out))
        assert out.tolist() == [fn(i) for i in range(out.size)], (dim, kind)
        outf = np.zeros(out.size, np.float32)
        L.orc_reduce_f32(ob.fp(inp.astype(np.float32)), _sz(outer), _sz(dimen), _sz(inner), kind, ob.fp(outf))
        assert outf.tolist() == [float(fn(i)) for i in range(out.size)]


def test_graph_assembled_from_primitive_ops_equals_net_forward():
    """The oracle's whole-net forward (hard-wired op schedule) against the same net put together op by op the way
    ModelIR::lower + the rewrite rules do: SparseMatmul per perspective, bias as
    Broadcast + add, SCReLU, concat as pad + pad + add (model/operations/index.rs:74-81), Matmul, bias,
    Select by output bucket, sigmoid, squared error, reduce-sum over the batch."""
    import os
    from bullet_amd import lib
    rec = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "batch.bf"), dtype=np.uint8)[:32 * 257]
    stm, ntm, bk, tg, w = lib.map_batch(lib.make_input_desc("chess768"), rec, out_buckets=8, blend=0.3, threads=1)
    B, H, NB = 257, 48, 8
    cfg = dict(num_inputs=768, nnz=32, ft_out=H, ft_act="screlu", num_buckets=NB, dense=[(1, "identity", True)])
    d = ob.make_net(**cfg)
    params = ob.net_init(d, 7)
    params[1][:] = np.linspace(-0.2, 0.3, H, dtype=np.float32)  # non-zero biases
    params[3][:] = np.linspace(-0.1, 0.1, NB, dtype=np.float32)
    l0w, l0b, l1w, l1b = params
    out_ref = np.zeros(B, np.float32)
    loss_ref = L.orc_net_forward(C.byref(d), B, ob.ptr_array(params), ob.ip(stm), ob.ip(ntm), ob.ip(bk), ob.fp(tg),
                                 ob.fp(w), ob.fp(out_ref))
    u = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
    hidden = []
    bias_b = np.zeros(B * H, np.float32)
    L.orc_broadcast(u(l0b), _sz(1), _sz(H), _sz(B), u(bias_b))          # [H] -> [B][H]
    for idx in (stm, ntm):
        acc = np.zeros(B * H, np.float32)
        L.orc_sparse_matmul_fwd(ob.fp(l0w), ob.ip(idx), ob.fp(acc), B, H, 768, H, 0, 32, 1)
        x = acc + bias_b                                                  # bias after the sum (builder.rs:145-147)
        y = np.zeros_like(x)
        L.orc_act_fwd(3, ob.fp(x), ob.fp(y), _sz(x.size))
        hidden.append(y)
    a_pad, b_pad = np.zeros(B * 2 * H, np.float32), np.zeros(B * 2 * H, np.float32)
    L.orc_pad(u(hidden[0]), _sz(B), _sz(H), _sz(1), _sz(0), _sz(H), C.c_uint32(0), u(a_pad))
    L.orc_pad(u(hidden[1]), _sz(B), _sz(H), _sz(1), _sz(H), _sz(0), C.c_uint32(0), u(b_pad))
    h = a_pad + b_pad                                                     # concat
    y_all = np.zeros(B * NB, np.float32)
    L.orc_matmul(ob.fp(l1w), ob.fp(h), ob.fp(y_all), 1, NB, 2 * H, B, 1, 1)  # W (NB x 2H) . X (2H x B), column-major
    b1 = np.zeros(B * NB, np.float32)
    L.orc_broadcast(u(l1b), _sz(1), _sz(NB), _sz(B), u(b1))
    y_all = y_all + b1
    y = np.zeros(B, np.float32)
    L.orc_select(ob.fp(y_all), ob.ip(bk), ob.fp(y), B, NB, NB)
    assert np.allclose(y, out_ref, rtol=2e-6, atol=1e-6)
    s = np.array([L.orc_sigmoid(C.c_float(v)) for v in y], dtype=np.float32)
    per_pos = (s - tg.reshape(-1)) ** 2
    total = np.zeros(1, np.float32)
    L.orc_reduce_f32(ob.fp(per_pos.astype(np.float32)), _sz(1), _sz(B), _sz(1), 0, ob.fp(total))
    assert abs(float(total[0]) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref))
