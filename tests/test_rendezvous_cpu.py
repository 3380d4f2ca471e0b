"""CPU: the file rendezvous of mfuoco_comm_create (c-lwe-snarks_amd/host/mfuoco_rendezvous.c; ADVICE round 3: a stale or planted id file must never send a
rank into ncclCommInitRank with an id the others do not have).  host/test_rendezvous compiles that file alone -- no GPU, no RCCL -- one process per rank."""
import os
import struct
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "c-lwe-snarks_amd", "host", "test_rendezvous")
MAGIC = 0x6D66756F636F4944


def _need():
    if not os.path.exists(EXE):
        pytest.fail("host/test_rendezvous has not been built (make -C c-lwe-snarks_amd dist)")


def _run(world, idf, tag, limit="30", delays=None, ranks=None):
    procs = {}
    for rk in (ranks if ranks is not None else range(world)):
        if delays and delays.get(rk):
            time.sleep(delays[rk])
        procs[rk] = subprocess.Popen([EXE, str(rk), str(world), idf, limit, tag], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return {rk: (p.communicate(timeout=120), p.returncode) for rk, p in procs.items()}


def _plant_stale(idf, world, nonce=0x1111222233334444, tag=b"S"):
    """what a killed earlier job leaves behind: its id file, its go file and the acks of its ranks (all well-formed, all with the OLD nonce)"""
    with open(idf, "wb") as f:
        f.write(struct.pack("<QQII", MAGIC, nonce, world, 0) + tag * 128)
    with open(idf + ".go", "wb") as f:
        f.write(struct.pack("<QQ", MAGIC, nonce) + struct.pack("<64Q", *([0] + [0xAAAA + k for k in range(1, 64)])))
    for k in range(1, world):
        with open(f"{idf}.ack.{k}", "wb") as f:
            f.write(struct.pack("<QQQ", MAGIC, nonce, 0xAAAA + k))


@pytest.mark.parametrize("world", [2, 3, 5, 8])
def test_ranks_agree_on_the_fresh_id(tmp_path, world):
    _need()
    idf = str(tmp_path / "comm_id")
    res = _run(world, idf, "A")
    for rk in range(world):
        (so, se), rc = res[rk]
        assert rc == 0, (rk, so, se)
        assert so.strip() == f"rank {rk} id " + "41" * 8


@pytest.mark.parametrize("world,late_rank0", [(3, False), (3, True), (8, False), (8, True)])
def test_stale_files_of_a_killed_job_are_not_believed(tmp_path, world, late_rank0):
    """id, go and ack files of an earlier session lie at the path; with late_rank0 the other ranks start FIRST and read the stale id before rank 0 has
    replaced it (they acknowledge the old nonce; rank 0 deletes those acks; they read again).  Every rank must end up with the NEW id.  world = 8: the rank count of the
    8 x MI355X node the C entry points are written for."""
    _need()
    idf = str(tmp_path / "comm_id")
    _plant_stale(idf, world)
    if late_rank0:
        procs = {rk: subprocess.Popen([EXE, str(rk), str(world), idf, "30", "N"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in range(1, world)}
        time.sleep(0.5)
        procs[0] = subprocess.Popen([EXE, "0", str(world), idf, "30", "N"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        res = {rk: (p.communicate(timeout=120), p.returncode) for rk, p in procs.items()}
    else:
        res = _run(world, idf, "N")
    for rk in range(world):
        (so, se), rc = res[rk]
        assert rc == 0, (rk, so, se)
        assert so.strip() == f"rank {rk} id " + "4e" * 8, "a rank left the rendezvous with the stale id"


def test_a_rank_alone_gives_up_with_an_error(tmp_path):
    """no rank 0 (it died, or never started): the others return -1 after the limit instead of waiting forever, even with a stale id + go at the path"""
    _need()
    idf = str(tmp_path / "comm_id")
    _plant_stale(idf, 2)
    t0 = time.time()
    res = _run(2, idf, "X", limit="1.5", ranks=[1])
    (so, se), rc = res[1]
    assert rc == 3 and "no rendezvous" in se and time.time() - t0 < 20
    assert not os.path.exists(idf + ".ack.1")  # its acknowledgement is withdrawn


def test_foreign_and_malformed_files_are_ignored(tmp_path):
    """a symlink at the ack path, a truncated id file: never followed, never parsed"""
    _need()
    idf = str(tmp_path / "comm_id")
    target = tmp_path / "victim"
    target.write_bytes(b"do not touch")
    os.symlink(str(target), idf + ".ack.1")
    with open(idf, "wb") as f:
        f.write(b"short")
    res = _run(2, idf, "B")
    for rk in range(2):
        (so, se), rc = res[rk]
        assert rc == 0, (rk, so, se)
        assert so.strip() == f"rank {rk} id " + "42" * 8
    assert target.read_bytes() == b"do not touch"
    st = os.stat(idf)
    assert (st.st_mode & 0o777) == 0o600
