"""The N > 1 host sequences on their REAL backend, as far as one GPU allows.

  * `dist.prove_sharded` / `dist.prove_batch_sharded` end to end under backend "nccl" (= RCCL) with a one-rank communicator in a fresh
    child process, `force_collectives=True`: librccl is loaded, a communicator is created on the MI355X and int64 device tensors go
    through all_reduce, all_to_all_single (split lists) and reduce_scatter_tensor; proofs must equal mfh_prove / mfh_prove_batch bit
    for bit (which test_gpu_batch_sizes.py pins to the oracle).  Reference loops that shard: src/snark.c:147-155, :157-174.
  * `python bench.py --gpus 2` with no launcher starts its own two ranks (here both on the one GPU, gloo host-staged collectives) and
    prints ONE JSON line with n_gpus = 2; a rank-count mismatch under an external launcher is an error, not a note.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


@pytest.mark.gpu
def test_sharded_sequences_under_one_rank_rccl(tmp_path):
    out = tmp_path / "rccl.txt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_child.py"), str(out), str(_free_port())], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert out.read_text().startswith("ok"), out.read_text()


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(tmp_path):
    """plain `python bench.py --gpus 2`: the parent only spawns, two fresh ranks share cuda:0 (gloo rehearsal backend)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "62", "--sharded-batch", "62",
                        "--no-resident", "--no-cpu-baseline"], env=_clean_env(MFUOCO_DIST_BACKEND="gloo", MFUOCO_SHARE_GPU="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["ranks"] == 2 and js["launcher"] == "self" and js["backend"].startswith("gloo")
    assert js["proof_accepted"] is True
    sb = js["row_sharded_batch"]
    assert sb["ranks"] == 2 and sb["own_proofs_accepted_rejected_as_expected_and_identical_to_prover"] is True
    ran = js["collectives"]["data_path_calls_on_rank0_all_legs"]
    assert ran["all_reduce"]["calls"] > 0 and ran["all_to_all_single"]["calls"] > 0 and ran["reduce_scatter_tensor"]["calls"] > 0


def test_bench_rank_count_mismatch_is_an_error():
    """an external launcher that started a different number of ranks than --gpus: exit code 2 before anything touches a GPU"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_launcher_propagates_a_failing_rank():
    """no GPU here: both self-started ranks fail at context creation; the parent (which never imports torch) returns non-zero and prints no JSON"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher's failure path")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rank process" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
