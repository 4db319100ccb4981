"""The C++ host API (include/bullet_amd.hpp), the mirror of bullet_lib's ValueTrainerBuilder / ValueTrainer.
CPU part: the graph closures of the reference's examples must lower to the net descriptors the kernels are
parity-tested with (tests/cpp/lower_nets.cpp, host code only). GPU part: examples/tests/test1.cpp and test2.cpp --
the reference's examples/tests/test1.rs / test2.rs rewritten against the mirror -- must reach the losses the reference
prints (0.056420 / 0.052189) and write the checkpoint tree of save_to_checkpoint."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "batch.bf")

EXPECTED = {
    "simple": "num_inputs=768 nnz=32 ft_out=128 ft_act=3 pairwise=0 perspectives=2 num_buckets=1 n_dense=1 "
              "L0=1/0/0/0/0 ft_factoriser=0 ft_segment=0 loss_kind=0 loss_power=2 eff=0",
    "output_buckets": "num_inputs=768 nnz=32 ft_out=1024 ft_act=3 pairwise=0 perspectives=2 num_buckets=8 n_dense=1 "
                      "L0=1/0/1/0/0 ft_factoriser=0 ft_segment=0 loss_kind=0 loss_power=2 eff=0",
    "multi_layer": "num_inputs=7680 nnz=32 ft_out=1024 ft_act=2 pairwise=1 perspectives=2 num_buckets=8 n_dense=3 "
                   "L0=16/3/1/0/0 L1=32/3/1/0/0 L2=1/0/1/0/0 ft_factoriser=10 ft_segment=768 loss_kind=0 loss_power=2 "
                   "eff=32",
    "pairwise_mul": "num_inputs=768 nnz=32 ft_out=64 ft_act=2 pairwise=1 perspectives=2 num_buckets=1 n_dense=1 "
                    "L0=1/0/0/0/0 ft_factoriser=0 ft_segment=0 loss_kind=1 loss_power=2.5 eff=0",
    "test2": "num_inputs=1536 nnz=32 ft_out=128 ft_act=2 pairwise=1 perspectives=2 num_buckets=8 n_dense=3 "
             "L0=16/2/1/1/1 L1=32/3/1/0/0 L2=1/0/1/0/0 ft_factoriser=2 ft_segment=768 loss_kind=0 loss_power=2 eff=0",
    "single_wdl": "num_inputs=147 nnz=49 ft_out=256 ft_act=3 pairwise=0 perspectives=1 num_buckets=1 n_dense=1 "
                  "L0=3/0/0/0/0 ft_factoriser=0 ft_segment=0 loss_kind=2 loss_power=0 eff=0",
}


def test_example_graphs_lower_to_the_tested_descriptors(tmp_path):
    exe = str(tmp_path / "lower_nets")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "lower_nets.cpp"), "-L" + os.path.join(ROOT, "bullet_amd"),
                           "-lbullet_amd", "-Wl,-rpath," + os.path.join(ROOT, "bullet_amd"), "-o", exe])
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    got = dict(line.split(" ", 1) for line in out if " " in line)
    sched = [float(x) for x in got.pop("sched").replace("|", " ").split()]
    f32 = np.float32
    lin = lambda sb: float(f32(0.001) + f32(0.0001 - 0.001) * f32(sb / 10.0))  # LinearDecayLR re-based after superbatch 4
    want = [0.001 / 10, 0.001, 0.001, lin(1), 0.0001, 0.3, 0.2, 0.0]  # LinearWDL over max - midpoint = 10 superbatches
    assert np.allclose(sched, want, rtol=2e-6, atol=1e-9), (sched, want)
    assert got == EXPECTED
    assert out[-1] == "rejected=3"  # graphs outside the family raise instead of being approximated


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.gpu
@pytest.mark.parametrize("name,expected,extra", [("test1", "0.056420", []), ("test2", "0.052189", []),
                                                 ("test1", "0.044886", ["128", "0.044886"])])  # = examples/tests/test3.rs
def test_cpp_examples_reach_the_reference_losses(tmp_path, name, expected, extra):
    if not _gpu():
        pytest.skip("needs a GPU")
    exe = os.path.join(ROOT, "examples", "bin", name)
    assert os.path.exists(exe), "examples/bin is built by __graft_entry__.build()"
    r = subprocess.run([exe, GOLDEN, str(tmp_path)] + extra, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"Expected loss: {expected}" in r.stdout and "Eval: " in r.stdout
    final = [ln for ln in r.stdout.splitlines() if ln.startswith("Final loss: ")][0].split()[-1]
    assert abs(float(final) - float(expected)) <= 1.5e-6
    ck = tmp_path / f"{name}-10"
    for f in ("optimiser_state/weights.bin", "optimiser_state/momentum.bin", "optimiser_state/velocity.bin", "raw.bin",
              "quantised.bin", "log.txt"):
        assert (ck / f).exists(), f
    log = (ck / "log.txt").read_text().strip().splitlines()
    assert [ln.split(",")[:2] for ln in log] == [[str(sb), "0"] for sb in range(1, 11)]  # value.rs:186-191
    if name == "test1" and not extra:  # l0w i16 + l0b i16 + l1w i16 + l1b i16, padded to 64 bytes (weights.rs:160-177)
        size = (768 * 32 + 32 + 64 + 1) * 2
        assert (ck / "quantised.bin").stat().st_size == (size + 63) // 64 * 64
        assert (ck / "raw.bin").stat().st_size == (768 * 32 + 32 + 64 + 1) * 4
        # trainer.optimiser.write_to_checkpoint / save_quantised / save_unquantised after the run == the checkpoint's files
        for a, b in (("manual_quantised.bin", "quantised.bin"), ("manual_raw.bin", "raw.bin"),
                     ("manual_state/weights.bin", "optimiser_state/weights.bin"),
                     ("manual_state/velocity.bin", "optimiser_state/velocity.bin")):
            assert (tmp_path / a).read_bytes() == (ck / b).read_bytes(), a


@pytest.mark.gpu
def test_cpp_multi_layer_example_trains_and_saves(tmp_path):
    """examples/progression/4_multi_layer.cpp (= the reference's 4_multi_layer.rs) for one short superbatch: factoriser,
    per-weight clipping, cosine LR, and the mixed i16 / i8 / f32 saved format with transposes and the factoriser merge."""
    if not _gpu():
        pytest.skip("needs a GPU")
    exe = os.path.join(ROOT, "examples", "bin", "4_multi_layer")
    r = subprocess.run([exe, GOLDEN, str(tmp_path), "1", "3"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    loss = float([ln for ln in r.stdout.splitlines() if ln.startswith("Final loss: ")][0].split()[-1])
    assert 0.01 < loss < 0.3
    ck = tmp_path / "4_multi_layer-1"
    nb, hl = 10, 1024
    size = 768 * nb * hl * 2 + hl * 2 + hl * 128 + 128 * 4 + 16 * 256 * 4 + 256 * 4 + 32 * 8 * 4 + 8 * 4
    assert (ck / "quantised.bin").stat().st_size == (size + 63) // 64 * 64
    assert (ck / "raw.bin").stat().st_size == (768 * nb * hl + hl + hl * 128 + 128 + 16 * 256 + 256 + 32 * 8 + 8) * 4
