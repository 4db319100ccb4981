"""Pins the CPU oracle against the reference's own e2e anchors:
examples/tests/test1.rs:54 / test4.rs (0.056420) and examples/tests/test3.rs:45 (0.044886):
10 AdamW steps (lr 1e-3, wdl 0.2, scale 400) over examples/tests/batch.bf (committed as the
data fixture tests/golden/batch.bf), batch 16384, weights from ModelWeights::new(seed 198273612).
The printed value is the loss of the 10th batch before its update (run.rs:212-221)."""
import os

import numpy as np
import pytest

import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "batch.bf")


def run_reference_schedule(hidden, steps=10, batch=16384):
    rec = np.fromfile(GOLDEN, dtype=np.uint8).reshape(-1, 32)
    n = rec.shape[0]
    assert n == 16273
    d = ob.make_net(768, 32, hidden, "screlu", [(1, "identity", False)])
    params = ob.net_init(d)
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    idesc = ob.input_desc("chess768")
    pos, losses = 0, []
    for _ in range(steps):
        # FixedSizeDataReader loops the file forever; ReadMapLoader re-chunks into exact batches
        # (crates/trainer/src/reader/fixed_size.rs:73-106, reader.rs:38-75)
        sel = np.arange(pos, pos + batch) % n
        pos += batch
        stm, ntm, bk, tg, _ = ob.map_batch(idesc, rec[sel].reshape(-1).copy(), blend=0.2)
        loss, grads, _ = ob.net_fwd_bwd(d, params, stm, ntm, bk, tg)
        for p_, m_, v_, g_ in zip(params, m, v, grads):
            ob.adamw(p_, m_, v_, g_, 1.0 / batch, 0.001)
        losses.append(loss / batch)
    return losses


@pytest.mark.parametrize("hidden,expected", [(32, "0.056420"), (128, "0.044886")])
def test_oracle_reproduces_reference_printed_loss(hidden, expected):
    losses = run_reference_schedule(hidden)
    assert "%.6f" % losses[-1] == expected
