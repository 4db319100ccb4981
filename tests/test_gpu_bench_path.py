"""The exact path bench.py times, against the oracle: bench.synth_records at full occupancy (32 pieces per position,
batch 16384) -> records uploaded -> bt_map_batch_device -> bt_net_hint_next_batch + bt_net_train_step, three steps over
three rotating device sets on a library stream, for BASELINE configs[1] (cfg2) and configs[2] (cfg3, the HalfKA net).
Losses of every step and all parameters after the last one are compared with the oracle's training step
(oracle/bullet_oracle.c: forward + backward + AdamW) on the host-mapped batches of the same records."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import bench  # noqa: E402
import gpu_util as gu  # noqa: E402
import oracle_binding as ob  # noqa: E402
from bullet_amd import lib  # noqa: E402

pytestmark = pytest.mark.gpu
L = lib.load()
B, STEPS = 16384, 3


@pytest.mark.parametrize("net_name", ["cfg2", "cfg3"])
def test_timed_path_matches_oracle(net_name):
    netcfg, input_kind, layout, _ = bench.NETS[net_name]
    idesc = lib.make_input_desc(input_kind, False, layout)
    nnz = netcfg["nnz"]
    recs = [bench.synth_records(B, 0xB011E7000 + 2 + 1000 * i, "full") for i in range(STEPS)]
    host = [lib.map_batch(idesc, r, out_buckets=8, blend=0.75, eval_scale=400.0, threads=8) for r in recs]
    assert all((h[0] >= 0).all() for h in host)  # full occupancy: every list has its 32 features

    # ---- device: the bench's step, on a library stream, three device sets, next batch announced
    net = lib.Net(lib.make_net_desc(**netcfg), B)
    ocfg = {k: v for k, v in netcfg.items() if k != "ft_segment"}  # (the segment promise is a device-side layout hint)
    do = ob.make_net(**ocfg)
    params = ob.net_init(do, ft_eff=32 if net_name != "cfg2" else 0)
    net.set_params(params)
    words = B * (2 * nnz + 3)
    packs = [torch.empty(words, dtype=torch.int32, device="cuda") for _ in range(STEPS)]

    def views(pack):
        o, out = 0, []
        for n, dt in ((B * nnz, torch.int32), (B * nnz, torch.int32), (B, torch.int32), (B, torch.float32), (B, torch.float32)):
            out.append(pack[o:o + n].view(dt).view(B, nnz) if n == B * nnz else pack[o:o + n].view(dt))
            o += n
        return out

    batches = [net.make_batch(*views(p)) for p in packs]
    ptrs = [[t.data_ptr() for t in views(p)] for p in packs]
    drecs = [gu.dev(r) for r in recs]
    mopts = lib.MapOpts(8, 0.75, 400.0, 0, 0, 1)
    sh = C.c_void_p()
    lib.check(L.bt_stream_create(C.byref(sh)))
    adj, lr = torch.tensor([1.0 / B], device="cuda"), torch.tensor([1e-3], device="cuda")
    torch.cuda.synchronize()
    for i in range(STEPS):  # (mapped on the step's own stream: no event needed for the hint)
        lib.check(L.bt_map_batch_device(sh, C.byref(idesc), C.byref(mopts), drecs[i].data_ptr(), B, *ptrs[i]))
    dev_loss = []
    losses = torch.zeros(STEPS, device="cuda")
    for i in range(STEPS):
        if i + 1 < STEPS:
            net.hint_next(batches[i + 1], None)
        check = L.bt_net_train_step(net.h, sh.value, C.byref(batches[i]), losses[i:].data_ptr(), adj.data_ptr(),
                                    lr.data_ptr(), None)
        lib.check(check)
    lib.check(L.bt_stream_sync(sh))
    torch.cuda.synchronize()
    dev_loss = [x / B for x in losses.cpu().tolist()]
    dev_params = net.get_tensors(net.params)
    # the device mapper filled the tensors the step consumed: bit-identical indices / buckets, targets within 1 ulp
    for i in range(STEPS):
        stm, ntm, bk, tg, w = (t.cpu().numpy() for t in views(packs[i]))
        assert np.array_equal(stm, host[i][0]) and np.array_equal(ntm, host[i][1]) and np.array_equal(bk, host[i][2])
        np.testing.assert_array_max_ulp(tg, host[i][3], maxulp=1)
    net.close()
    L.bt_stream_destroy(sh)

    # ---- oracle: the same three steps
    m, v = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
    orc_loss = []
    for i in range(STEPS):
        stm, ntm, bk, tg, w = host[i]
        loss, grads, _ = ob.net_fwd_bwd(do, params, stm, ntm, bk, tg, w)
        orc_loss.append(loss / B)
        for p_, m_, v_, g_ in zip(params, m, v, grads):
            ob.adamw(p_, m_, v_, g_, 1.0 / B, 1e-3)
    # fp32 bars of the other whole-net tests (tests/test_gpu_net.py): loss 1e-5 relative, parameters 1e-4 relative
    for a, b in zip(dev_loss, orc_loss):
        assert abs(a - b) <= 1e-5 * abs(b) + 1e-7, (dev_loss, orc_loss)
    for t, (a, b) in enumerate(zip(dev_params, params)):
        gu.assert_close(a, b, rtol=1e-4, atol=2e-5, what=f"{net_name}: tensor {t} after {STEPS} timed-path steps")
