"""Helpers for the -m gpu parity tests: run ops through the C ABI on torch-owned device memory."""
import ctypes as C

import numpy as np
import torch

import oracle_binding as ob
from bullet_amd import lib

L = lib.load()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stream():
    return torch.cuda.current_stream().cuda_stream


def random_indices(rng, batch, nnz, cols, min_cnt=0, stale_tail=True, allow_dups=False):
    """[batch][nnz] index lists: valid prefix, -1 terminator, then stale garbage (as the reference's
    recycled host buffers leave it)."""
    idx = np.empty((batch, nnz), dtype=np.int32)
    for b in range(batch):
        cnt = int(rng.integers(min_cnt, nnz + 1))
        if allow_dups:
            feats = rng.integers(0, cols, size=cnt)
        else:
            feats = rng.choice(cols, size=min(cnt, cols), replace=False)
            cnt = feats.size
        idx[b, :cnt] = feats
        if cnt < nnz:
            idx[b, cnt] = -1
            tail = rng.integers(-1, cols, size=nnz - cnt - 1) if stale_tail else -np.ones(nnz - cnt - 1)
            idx[b, cnt + 1:] = tail
    return idx


def ft_forward(desc, w, bias, idxs, pairwise_saved=False):
    out_w = desc.rows // 2 if desc.pairwise else desc.rows
    out = torch.full((desc.batch, desc.out_ld), float("nan"), device="cuda")
    saved = torch.full((desc.n_inputs, desc.batch, desc.rows), float("nan"), device="cuda") if desc.pairwise else None
    dw, db = dev(w), (dev(bias) if bias is not None else None)
    di = [dev(i) for i in idxs]
    ptrs = (C.c_void_p * 2)(*[t.data_ptr() for t in di] + [None] * (2 - len(di)))
    if desc.segment > 0:  # segment-grouped tables: entry lists first, then the forward that walks them
        wsb = L.bt_ft_backward_workspace_bytes(C.byref(desc))
        assert wsb > 0, L.bt_last_error()
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        lib.check(L.bt_ft_build_index(stream(), C.byref(desc), ptrs, ws.data_ptr(), wsb))
        lib.check(L.bt_ft_forward_prebuilt(stream(), C.byref(desc), dw.data_ptr(),
                                           db.data_ptr() if db is not None else None, ptrs, out.data_ptr(),
                                           saved.data_ptr() if saved is not None else None, ws.data_ptr(), wsb))
        viol = C.c_int(-1)
        lib.check(L.bt_ft_segment_violations(stream(), C.byref(desc), ws.data_ptr(), wsb, C.byref(viol)))
        ft_forward.last_violations = viol.value
    else:
        lib.check(L.bt_ft_forward(stream(), C.byref(desc), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                  ptrs, out.data_ptr(), saved.data_ptr() if saved is not None else None))
    torch.cuda.synchronize()
    return out.cpu().numpy(), (saved.cpu().numpy() if saved is not None else None)


def segment_indices(rng, batch, nnz, nseg, seg_cols, seg_probs=None, min_cnt=0):
    """index lists whose valid prefix stays inside one segment per entry (king-bucketed input sets); the stale tail
    behind the -1 terminator may point anywhere"""
    idx = np.empty((batch, nnz), dtype=np.int32)
    segs = rng.choice(nseg, size=batch, p=seg_probs)
    for b in range(batch):
        cnt = int(rng.integers(min_cnt, nnz + 1))
        idx[b, :cnt] = segs[b] * seg_cols + rng.choice(seg_cols, size=cnt, replace=False)
        if cnt < nnz:
            idx[b, cnt] = -1
            idx[b, cnt + 1:] = rng.integers(-1, nseg * seg_cols, size=nnz - cnt - 1)
    return idx


def ft_backward(desc, idxs, grad_out, out, saved, want_bias=True):
    di = [dev(i) for i in idxs]
    ptrs = (C.c_void_p * 2)(*[t.data_ptr() for t in di] + [None] * (2 - len(di)))
    wsb = L.bt_ft_backward_workspace_bytes(C.byref(desc))
    assert wsb > 0, L.bt_last_error()
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dw = torch.full((desc.stride * desc.cols,), float("nan"), device="cuda")
    dbias = torch.full((desc.rows,), float("nan"), device="cuda") if want_bias else None
    go, o = dev(grad_out), dev(out)
    sv = dev(saved) if saved is not None else None
    lib.check(L.bt_ft_backward(stream(), C.byref(desc), ptrs, go.data_ptr(), o.data_ptr(),
                               sv.data_ptr() if sv is not None else None, dw.data_ptr(),
                               dbias.data_ptr() if dbias is not None else None, ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    return dw.cpu().numpy(), (dbias.cpu().numpy() if dbias is not None else None)


def oracle_ft_forward(desc, w, bias, idxs):
    """oracle composition of SparseMatmul(break rule) + bias + act (+ pairwise)"""
    O = ob.lib()
    outs, saveds = [], []
    for idx in idxs:
        acc = np.zeros((desc.batch, desc.rows), np.float32)
        O.orc_sparse_matmul_fwd(ob.fp(w), ob.ip(idx), ob.fp(acc), desc.batch, desc.rows, desc.cols, desc.stride,
                                desc.offset, desc.nnz, 1)
        if bias is not None:
            acc = (acc + bias[None, :]).astype(np.float32)
        c = np.zeros_like(acc)
        O.orc_act_fwd(desc.act, ob.fp(acc), ob.fp(c), C.c_size_t(acc.size))
        if desc.pairwise:
            h = desc.rows // 2
            outs.append((c[:, :h] * c[:, h:]).astype(np.float32))
            saveds.append(c)
        else:
            outs.append(c)
    out = np.concatenate(outs, axis=1)
    return out, (np.stack(saveds) if desc.pairwise else None)


def oracle_ft_backward(desc, idxs, grad_out, out, saved):
    O = ob.lib()
    out_w = desc.rows // 2 if desc.pairwise else desc.rows
    gs = []
    for k in range(len(idxs)):
        go = np.ascontiguousarray(grad_out[:, k * out_w:(k + 1) * out_w])
        if desc.pairwise:
            c = saved[k]
            h = out_w
            d0, d1 = np.zeros((desc.batch, h), np.float32), np.zeros((desc.batch, h), np.float32)
            c0, c1 = np.ascontiguousarray(c[:, :h]), np.ascontiguousarray(c[:, h:])
            O.orc_act_bwd(desc.act, ob.fp(c0), ob.fp((go * c1).astype(np.float32)), ob.fp(d0), C.c_size_t(d0.size))
            O.orc_act_bwd(desc.act, ob.fp(c1), ob.fp((go * c0).astype(np.float32)), ob.fp(d1), C.c_size_t(d1.size))
            g = np.concatenate([d0, d1], axis=1)
        else:
            o = np.ascontiguousarray(out[:, k * out_w:(k + 1) * out_w])
            g = np.zeros_like(o)
            O.orc_act_bwd(desc.act, ob.fp(o), ob.fp(go), ob.fp(g), C.c_size_t(g.size))
        gs.append(np.ascontiguousarray(g))
    n = len(idxs)
    gp = (C.POINTER(C.c_float) * n)(*[ob.fp(g) for g in gs])
    ipp = (C.POINTER(C.c_int32) * n)(*[ob.ip(i) for i in idxs])
    rows, offs = (C.c_int * n)(*[desc.rows] * n), (C.c_int * n)(*[desc.offset] * n)
    dw = np.zeros(desc.stride * desc.cols, np.float32)
    O.orc_sparse_matmul_bwd_multi(n, gp, ipp, ob.fp(dw), desc.batch, rows, desc.cols, desc.stride, offs, desc.nnz, 1)
    dbias = np.zeros(desc.rows, np.float32)
    for g in gs:
        t = np.zeros(desc.rows, np.float32)
        O.orc_reduce_sum_batch(ob.fp(g), ob.fp(t), desc.batch, desc.rows)
        dbias += t
    # magnitude of the summed terms, for a summation-order-aware tolerance
    return dw, dbias, gs


def assert_close(a, b, rtol, atol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} mismatches, max err {err.max():.3e} " \
                          f"(at ref {b.reshape(-1)[err.argmax()]:.4e})"
