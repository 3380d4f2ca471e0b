/*
 * mfuoco_dist.h -- multi-GPU entry points of the host shim (libmfuoco_gpu_dist.so): one process per GPU, RCCL over xGMI called
 * directly from C.  Not in the reference (it is single-threaded, single-device); these are what a maintainer of
 * src/benchmark_snark.c:70-74 calls instead of prover() to use more than one GPU.  The loops that shard are src/snark.c:147-155
 * (b_w over the BT+BV rows) and :157-174 (the four eval_poly passes over the S / AS rows): rank r owns rows [R r / world, R (r+1) / world)
 * of every CRS region; SURVEY 8(e).
 *
 * Link:  -lmfuoco_gpu_dist -lmfuoco_gpu -lmfhip -lrccl -lgmp        
 */
#ifndef MFUOCO_DIST_H
#define MFUOCO_DIST_H
#include "mfuoco/mangiafuoco_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfuoco_comm mfuoco_comm;

#define MFUOCO_UNIQUE_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

/* Communicator of `world` processes, this one being `rank`, on GPU `device` (which also becomes the shim's device: call before any other
 * shim function).  The 128-byte ncclUniqueId travels through the file `id_file`: rank 0 creates it (ncclGetUniqueId), writes it under a
 * temporary name and renames it into place; the other ranks wait for the file (at most 120 s).  Remove the file between jobs.
 * Returns 0, or -1 with a message on stderr. */
int mfuoco_comm_create(mfuoco_comm **comm, int rank, int world, int device, const char *id_file);
/* the same with the id handed over by the caller's own launcher (MPI_Bcast, a socket, an environment variable ...) */
void mfuoco_comm_unique_id(uint8_t id[MFUOCO_UNIQUE_ID_BYTES]);
int mfuoco_comm_create_from_id(mfuoco_comm **comm, int rank, int world, int device, const uint8_t id[MFUOCO_UNIQUE_ID_BYTES]);
/* REHEARSAL backend for boxes with fewer GPUs than ranks (tests): the same call sequence with every collective staged through a POSIX
 * shared-memory segment `/name` on the host (what gloo is to the Python driver); all ranks may share one GPU.  Never the product path. */
int mfuoco_comm_create_rehearsal(mfuoco_comm **comm, int rank, int world, int device, const char *shm_name);
void mfuoco_comm_destroy(mfuoco_comm *comm);
int mfuoco_comm_rank(const mfuoco_comm *comm);
int mfuoco_comm_world(const mfuoco_comm *comm);
/* "rccl" or "rehearsal (host shared memory)" */
const char *mfuoco_comm_backend(const mfuoco_comm *comm);
/* calls and bytes handed to the backend so far: index 0 all-to-all (send/recv groups), 1 reduce-scatter, 2 all-reduce, 3 broadcast */
void mfuoco_comm_stats(const mfuoco_comm *comm, uint64_t calls[4], uint64_t bytes[4]);

/* prover() (src/snark.h:50, src/snark.c:117-190) for `count` statements under one CRS and SSP with the CRS ROWS sharded over the ranks
 * of `comm` -- what c-lwe-snarks_amd/dist.py:prove_batch_sharded does, in C:
 *   chain of the rank's own statement slab (mfh_batch_chain)  ->  all-to-all of the w | h | v row slices (ncclGroupStart + ncclSend / ncclRecv)
 *   ->  mfh_prove_batch_partial on the rank's row shares  ->  mfh_ct_to_lanes  ->  ONE ncclReduceScatter(sum, ncclUint64)
 *   ->  mfh_ct_from_lanes  ->  mfh_prove_batch_finish (delta ct_t, smudging) on the own slab.
 * Every rank passes the same crs, ssp, witnesses and count.  Statements are owned in slabs of ceil(count / world): on return
 * [*own_first, *own_first + *own_count) are this rank's statements and pis[k] (initialised by proof_init) holds their proofs; the other
 * pis[] are untouched.  Entropy (delta, the five smudging draws) is drawn by the owner, per statement in prover()'s order.  With the same
 * randomness every proof is bit-identical to prover()'s and mfuoco_prover_batch()'s.  comm == NULL or a one-rank communicator still runs
 * the whole sequence (through the backend when there is one). */
void mfuoco_prover_batch_sharded(proof_t *pis, crs_t crs, ssp_t ssp, mpz_t *witnesses, size_t count, mfuoco_comm *comm, size_t *own_first,
                                 size_t *own_count);
/* ONE proof computed by all ranks together (every rank returns the complete proof): the rank's share of the witness polynomial ->
 * ncclAllReduce(sum, ncclUint64) over D lanes -> the rank's row shares of the five ciphertexts -> ncclAllReduce over 5 x 1471 x 22 lanes ->
 * carries, modq, smudging.  Rank 0 draws the entropy and broadcasts it (ncclBroadcast, 413 bytes). */
void mfuoco_prover_sharded(proof_t pi, crs_t crs, ssp_t ssp, mpz_t witness, mfuoco_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
