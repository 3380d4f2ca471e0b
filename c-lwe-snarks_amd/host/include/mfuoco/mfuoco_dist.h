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

/* Communicator of `world` processes, this one being `rank`, on GPU `device`, which must be the shim's GPU: call this before any other shim function, or
 * after mfuoco_gpu_set_device(device); -1 if the shim already runs on another GPU.  The 128-byte ncclUniqueId travels through files named after
 * `id_file` (a path in a directory only this user can write -- the caller's job directory, not a shared /tmp): rank 0 removes what an earlier job may
 * have left, publishes the id with a fresh session nonce, collects an acknowledgement of THAT nonce from every rank and only then releases them, so that no
 * rank enters ncclCommInitRank with a stale or foreign id (host/mfuoco_dist.c, "rendezvous through files").  Every file is created exclusively, mode 0600,
 * and removed once the communicator exists.  Gives up after $MFUOCO_RENDEZVOUS_S seconds (default 120).  Returns 0, or -1 with a message on stderr: exit
 * non-zero then, do not retry inside the process. */
int mfuoco_comm_create(mfuoco_comm **comm, int rank, int world, int device, const char *id_file);
/* the same with the id handed over by the caller's own launcher (MPI_Bcast, a socket, an environment variable ...) */
void mfuoco_comm_unique_id(uint8_t id[MFUOCO_UNIQUE_ID_BYTES]);
int mfuoco_comm_create_from_id(mfuoco_comm **comm, int rank, int world, int device, const uint8_t id[MFUOCO_UNIQUE_ID_BYTES]);
/* A communicator over the caller's OWN collectives (an MPI build, a test harness) instead of RCCL: the four operations the sequences below need, on device
 * buffers, ordered ON `stream` (a hipStream_t; NULL = the NULL stream): the operation reads its inputs after the work queued on that stream before the call and
 * work queued on it after the call sees the outputs -- RCCL takes the stream as it is; a transport that stages through the host synchronises that stream itself.
 * The batch prover issues its collectives on a stream of the communicator's own, every rank in the same order.  Element counts, not bytes;
 * reduce_scatter: d_recv[0 .. n) = sum over ranks q of q's d_send[rank * n .. (rank + 1) * n), wrap-around uint64 sums; destroy may be NULL. */
typedef struct mfuoco_transport {
  const char *name;
  void (*alltoallv_u32)(void *impl, int rank, int world, const uint32_t *d_send, const size_t *scnt, const size_t *sdsp, uint32_t *d_recv, const size_t *rcnt,
                        const size_t *rdsp, void *stream);
  void (*reduce_scatter_u64)(void *impl, int rank, int world, const uint64_t *d_send, uint64_t *d_recv, size_t n, void *stream);
  void (*allreduce_u64)(void *impl, int rank, int world, uint64_t *d_buf, size_t n, void *stream);
  void (*bcast_bytes)(void *impl, int rank, int world, uint8_t *d_buf, size_t n, int root, void *stream);
  void (*destroy)(void *impl);
} mfuoco_transport;
int mfuoco_comm_create_transport(mfuoco_comm **comm, int rank, int world, int device, const mfuoco_transport *transport, void *impl);
void mfuoco_comm_destroy(mfuoco_comm *comm);
int mfuoco_comm_rank(const mfuoco_comm *comm);
int mfuoco_comm_world(const mfuoco_comm *comm);
/* the transport's name: "rccl" for the communicators this library creates itself */
const char *mfuoco_comm_backend(const mfuoco_comm *comm);
/* calls and bytes handed to the backend so far: index 0 all-to-all (send/recv groups), 1 reduce-scatter, 2 all-reduce, 3 broadcast */
void mfuoco_comm_stats(const mfuoco_comm *comm, uint64_t calls[4], uint64_t bytes[4]);

/* prover() (src/snark.h:50, src/snark.c:117-190) for `count` statements under one CRS and SSP with the CRS ROWS sharded over the ranks
 * of `comm` -- what c-lwe-snarks_amd/dist.py:prove_batch_sharded does, in C:
 *   chain of the rank's own statement slab (mfh_batch_chain)  ->  all-to-all of the w | h | v row slices (ncclGroupStart + ncclSend / ncclRecv)
 *   ->  mfh_prove_batch_partial on the rank's row shares  ->  mfh_ct_to_lanes  ->  ncclReduceScatter(sum, ncclUint64)
 *   ->  mfh_ct_from_lanes  ->  mfh_prove_batch_finish (delta ct_t, smudging) on the own slab,
 * pipelined in stages of (255 / world) statements per rank (at most one super-group of the row work): the collectives run on the communicator's own stream,
 * stage k + 1's all-to-all and stage k - 1's reduce-scatter, finish and device-to-host drain under the row work of stage k; the bytes handed to the backend are those
 * of the one-shot sequence ($MFUOCO_DIST_STAGE=0 restores it).  The cut is a function of count, the number of ranks and the environment, so every rank issues the same
 * collectives; a rank whose image share is not kept by the shim expands it once per call.
 * Every rank passes the same crs, ssp, witnesses and count.  Statements are owned in slabs of ceil(count / world): on return
 * [*own_first, *own_first + *own_count) are this rank's statements and pis[k] (initialised by proof_init) holds their proofs; the other
 * pis[] are untouched.  Entropy (delta, the five smudging draws) is drawn by the owner, per statement in prover()'s order.  With the same
 * randomness every proof is bit-identical to prover()'s and mfuoco_prover_batch()'s.  comm == NULL or a one-rank communicator still runs
 * the whole sequence (through the backend when there is one). */
void mfuoco_prover_batch_sharded(proof_t *pis, crs_t crs, ssp_t ssp, mpz_t *witnesses, size_t count, mfuoco_comm *comm, size_t *own_first,
                                 size_t *own_count);
/* ONE proof computed by all ranks together (every rank returns the complete proof): the rank's share of the witness polynomial ->
 * ncclAllReduce(sum, ncclUint64) over D lanes -> the rank's row shares of the five ciphertexts -> ncclAllReduce over 5 x 1471 x 13 lanes (56 bits per uint64 lane) ->
 * carries, modq, smudging.  Rank 0 draws the entropy and broadcasts it (ncclBroadcast, 413 bytes). */
void mfuoco_prover_sharded(proof_t pi, crs_t crs, ssp_t ssp, mpz_t witness, mfuoco_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
