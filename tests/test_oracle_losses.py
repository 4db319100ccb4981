"""The oracle's restatement of the loss functions beyond sigmoid-MSE and of the clip / decay kernels (SURVEY 8f row
4). The reference holds no vectors for these (no test or example uses them), so they are checked against an independent
float64 numpy statement of the published formulas and against finite differences: **parity unpinned** beyond that."""
import numpy as np
import pytest

import oracle_binding as ob


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _loss64(kind, y, t, w, power):
    y, t = y.astype(np.float64), t.astype(np.float64)
    if kind == "softmax_ce":
        z = y - y.max(axis=1, keepdims=True)
        logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
        per = -(t * logp).sum(axis=1)
    elif kind == "sigmoid_power":
        per = np.abs(_sigmoid(y[:, 0]) - t[:, 0]) ** power
    else:
        per = (_sigmoid(y[:, 0]) - t[:, 0]) ** 2
    return float((per * (w if w is not None else 1.0)).sum())


@pytest.mark.parametrize("kind,width,power", [("sigmoid_mse", 1, 0.0), ("sigmoid_power", 1, 2.5), ("sigmoid_power", 1, 3.0),
                                              ("softmax_ce", 3, 0.0), ("softmax_ce", 5, 0.0)])
@pytest.mark.parametrize("use_w", [False, True])
def test_loss_value_and_gradient(kind, width, power, use_w):
    rng = np.random.default_rng(width + int(power * 10) + use_w)
    B = 64
    y = (rng.standard_normal((B, width)) * 1.5).astype(np.float32)
    if kind == "softmax_ce":
        t = rng.dirichlet(np.ones(width), size=B).astype(np.float32)
        t[::2] = np.eye(width, dtype=np.float32)[rng.integers(0, width, t[::2].shape[0])]
    else:
        t = rng.random((B, width)).astype(np.float32)
    w = (rng.random(B) + 0.5).astype(np.float32) if use_w else None
    l, gy = ob.loss(kind, y, t, w, power)
    assert abs(l - _loss64(kind, y, t, w, power)) <= 2e-6 * B
    eps = 1e-4
    for (b, k) in [(0, 0), (7, width - 1), (63, 0)]:
        yp, ym = y.astype(np.float64).copy(), y.astype(np.float64).copy()
        yp[b, k] += eps
        ym[b, k] -= eps
        fd = (_loss64(kind, yp, t, w, power) - _loss64(kind, ym, t, w, power)) / (2 * eps)
        assert abs(gy[b, k] - fd) <= 1e-5 + 1e-4 * abs(fd)


def test_power_loss_gradient_is_zero_at_zero_difference():
    y = np.array([[0.3]], np.float32)
    t = np.array([[ob.lib().orc_sigmoid(0.3)]], np.float32)
    l, gy = ob.loss("sigmoid_power", y, t, None, 2.5)
    assert l == 0.0 and gy[0, 0] == 0.0  # sgn(0) = 0 (gpu/src/pointwise/write.rs:172-174)


def test_weight_clip_and_decay():
    w = np.array([-3.0, -1.98, 0.25, 1.98, 2.5], np.float32)
    ob.weight_clip(w, -1.98, 1.98)
    assert w.tolist() == [np.float32(-1.98), np.float32(-1.98), 0.25, np.float32(1.98), np.float32(1.98)]
    w = np.array([1.0, -2.0], np.float32)
    ob.weight_decay(w, 0.01, 0.5)
    f = np.float32(1.0) + np.float32(-1.0) * (np.float32(0.5) * np.float32(0.01))
    assert w.tolist() == [f, np.float32(-2.0) * f]


def test_clip_pass_through_and_faux_quantise():
    """the two leftover pointwise model ops (builder.rs:365-373). The reference holds no vectors for them (parity
    unpinned): the oracle is checked against the formulas written out in numpy float32."""
    x = np.array([-3.0, -1.98, -0.26, 0.0, 0.2549, 0.25, 1.98, 2.5], np.float32)
    assert ob.clip_pass_through(x, -1.98, 1.98).tolist() == np.clip(x, np.float32(-1.98), np.float32(1.98)).tolist()
    q = np.float32(255.0)
    scaled = q * x
    half_away = np.sign(scaled) * np.floor(np.abs(scaled) + np.float32(0.5))  # f32::round: ties away from zero
    assert ob.faux_quantise(x, 255.0, True).tolist() == (half_away * (np.float32(1.0) / q)).astype(np.float32).tolist()
    assert ob.faux_quantise(x, 255.0, False).tolist() == (np.trunc(scaled) * (np.float32(1.0) / q)).astype(np.float32).tolist()
    tie = np.array([0.5, -0.5, 1.5, 2.5], np.float32)  # value 1: round() must not be round-half-to-even
    assert ob.faux_quantise(tie, 1.0, True).tolist() == [1.0, -1.0, 2.0, 3.0]
    assert ob.faux_quantise(tie, 1.0, False).tolist() == [0.0, -0.0, 1.0, 2.0]


def test_wdl_net_oracle_gradient_matches_finite_differences():
    """3-output net with softmax-CE on one-hot targets: dL/d(bias of the head) by central differences"""
    rng = np.random.default_rng(0)
    d = ob.make_net(num_inputs=768, nnz=8, ft_out=16, ft_act="screlu", dense=[(3, "identity", False)], loss="softmax_ce")
    params = ob.net_init(d, 3)
    B = 32
    stm = rng.integers(0, 768, (B, 8)).astype(np.int32)
    ntm = rng.integers(0, 768, (B, 8)).astype(np.int32)
    bk = np.zeros(B, np.int32)
    tg = np.eye(3, dtype=np.float32)[rng.integers(0, 3, B)].reshape(-1).copy()
    l0, grads, out = ob.net_fwd_bwd(d, params, stm, ntm, bk, tg)
    assert out.shape == (3 * B,)
    for j in range(3):
        eps = 1e-2
        p1 = [p.copy() for p in params]
        p1[3][j] += eps
        p2 = [p.copy() for p in params]
        p2[3][j] -= eps
        fd = (ob.net_fwd_bwd(d, p1, stm, ntm, bk, tg)[0] - ob.net_fwd_bwd(d, p2, stm, ntm, bk, tg)[0]) / (2 * eps)
        assert abs(grads[3][j] - fd) <= 2e-3 * max(1.0, abs(fd))
