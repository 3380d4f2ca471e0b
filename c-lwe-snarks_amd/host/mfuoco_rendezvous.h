/*
 * mfuoco_rendezvous.h -- internal to libmfuoco_gpu_dist (hidden symbols): how the ranks of mfuoco_comm_create agree on ONE fresh 128-byte communicator id
 * through files, without trusting anything an earlier or foreign job left at that path.  Pure POSIX (no GPU): host/test_rendezvous.c exercises it on CPU.
 */
#ifndef MFUOCO_RENDEZVOUS_H
#define MFUOCO_RENDEZVOUS_H
#include <stdint.h>
#define MFUOCO_RDV_ID_BYTES 128
#define MFUOCO_RDV_MAXW 64
/* rank 0 passes its fresh id in `id`; on return 0 every rank holds rank 0's id of THIS session.  -1 (message on stderr) after limit_s seconds. */
__attribute__((visibility("hidden"))) int mfuoco_rendezvous_files(int rank, int world, const char *id_file, uint8_t id[MFUOCO_RDV_ID_BYTES], double limit_s);
/* removes the files this rank created (call once the communicator exists) */
__attribute__((visibility("hidden"))) void mfuoco_rendezvous_cleanup(int rank, const char *id_file);
#endif
