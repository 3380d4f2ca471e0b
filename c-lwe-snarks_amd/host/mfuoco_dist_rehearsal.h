/*
 * mfuoco_dist_rehearsal.h -- TEST SCAFFOLDING, not part of libmfuoco_gpu_dist.so: a transport (mfuoco_dist.h: mfuoco_transport) that stages every
 * collective through a POSIX shared-memory segment on the host, for boxes with fewer GPUs than ranks (what gloo is to the Python driver); all ranks may
 * share one GPU.  Compiled into host/test_sharded and host/bench_snark_sharded only (c-lwe-snarks_amd/Makefile, target dist).
 */
#ifndef MFUOCO_DIST_REHEARSAL_H
#define MFUOCO_DIST_REHEARSAL_H
#include "mfuoco/mfuoco_dist.h"

/* the same call sequence as mfuoco_comm_create with the segment `/shm_name` as the wire; $MFUOCO_REHEARSAL_SLOT_MB sizes a rank's mailbox (default 256) */
int mfuoco_comm_create_rehearsal(mfuoco_comm **comm, int rank, int world, int device, const char *shm_name);
#endif
