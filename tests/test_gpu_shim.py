"""GPU: the reference-signature C shim (libmfuoco_gpu_debug.so) driven by a C program that exercises what the shim adds to the reference's interface --
batch encryption / decryption / prover / verifier, the images kept across calls, the on-disk formats (c-lwe-snarks_amd/host/test_shim.c) -- and the C entry
points for N GPUs.  (The reference's own test programs run against the shim in test_gpu_reference_drivers.py.)"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_shim_properties():
    exe = os.path.join(ROOT, "c-lwe-snarks_amd", "host", "test_shim")
    if not os.path.exists(exe):
        pytest.fail("host/test_shim has not been built (make -C c-lwe-snarks_amd shim); the shim needs gmp.h at build time")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "encrypt batch ok" in r.stdout and "snark ok" in r.stdout and "files ok" in r.stdout


def _sharded_exe():
    exe = os.path.join(ROOT, "c-lwe-snarks_amd", "host", "test_sharded")
    if not os.path.exists(exe):
        pytest.fail("host/test_sharded has not been built (make -C c-lwe-snarks_amd dist)")
    return exe


def _rank_env(rank, world, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MFUOCO_REHEARSAL_SHM")}
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


@pytest.mark.parametrize("stage,resident", [(None, True), ("7", True), ("0", True), ("7", False)])
def test_c_sharded_prover_one_rank_rccl(tmp_path, stage, resident):
    """mfuoco_prover_batch_sharded / mfuoco_prover_sharded (host/mfuoco_dist.c) through librccl called from C: a one-rank communicator still
    sends the all-to-all (ncclSend/ncclRecv to itself), the ncclReduceScatter, both ncclAllReduce and the ncclBroadcast; proofs must equal
    mfuoco_prover_batch's / prover()'s bit for bit on the same entropy tape and verify.  stage ($MFUOCO_DIST_STAGE): statements per rank and pipeline stage --
    None: the plan of a real call (40 statements: one stage); 7: six stages (7, 7, 7, 7, 7, 5), the collectives of stage k + 1 / k - 1 on the communicator's stream
    beside the row work of stage k, every buffer re-used by parity three times; 0: the one-shot sequence.  resident = False ($MFUOCO_GPU_RESIDENT_CRS=0): the shim
    keeps no image share, the staged call expands its own once and drops it -- the cut of the call (hence the collectives every rank issues) must not depend on it"""
    extra = {"MFUOCO_DIST_STAGE": stage} if stage is not None else {}
    if not resident:
        extra["MFUOCO_GPU_RESIDENT_CRS"] = "0"
    r = subprocess.run([_sharded_exe(), "40"], env=_rank_env(0, 1, MFUOCO_COMM_ID_FILE=str(tmp_path / "id"), **extra), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "backend=rccl" in r.stdout and "sharded ok" in r.stdout
    nst = {None: 1, "7": 6, "0": 1}[stage]
    assert f"all_to_all={nst + 1} reduce_scatter={nst + 1}" in r.stdout, r.stdout  # (+ the second, one-statement call)


@pytest.mark.parametrize("world,count,stage", [(2, 9, None), (3, 40, None), (4, 41, None), (3, 40, "5"), (4, 41, "4"), (4, 9, "1")])
def test_c_sharded_prover_rehearsal_ranks_share_the_gpu(world, count, stage):
    """the same C sequence with `world` PROCESSES on the one GPU, collectives staged through host shared memory (rehearsal backend): uneven
    statement slabs and row shares (256 rows over 3 ranks), a second call in which the last rank owns no statement.  4 ranks beside this test runner (which holds the GPU itself) stay clear of what the GPU box lets one
    command put on its card at once (its process guard: 6 processes; 6 ranks + the runner were killed by it in round 5), so the machine's real rank count, 8, is rehearsed on the CPU only: the host sequence over gloo in
    tests/test_dist_cpu.py, the rendezvous in tests/test_rendezvous_cpu.py.
    stage ($MFUOCO_DIST_STAGE): several pipeline stages per call -- 40 statements over 3 ranks in stages of 5 per rank: slabs of 14, 14, 12 = stages of 15, 15, 10 (rank 2
    contributes 2 to the last); 41 over 4 ranks in stages of 4: slabs of 11, 11, 11, 8 = stages of 16, 16, 9 (rank 3 owns none of the last); 9 over 4 ranks in stages of 1: slabs of
    3, 3, 3, 0 -- a rank with NO statement walks stages whose first own index lies beyond its slab (the chain-pass loop spun there forever in the Python mirror before it was
    bounded by the slab: found by tests/test_dist_cpu.py at 8 ranks)."""
    name = "mfuoco_test_%d_%d" % (os.getpid(), world)
    extra = {"MFUOCO_DIST_STAGE": stage} if stage is not None else {}
    procs = [subprocess.Popen([_sharded_exe(), str(count)], env=_rank_env(rk, world, MFUOCO_REHEARSAL_SHM=name, MFUOCO_SHARE_GPU="1", **extra),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=600))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for rk, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0, (rk, so, se)
        assert "sharded ok" in so and "rehearsal" in so
