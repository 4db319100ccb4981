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
