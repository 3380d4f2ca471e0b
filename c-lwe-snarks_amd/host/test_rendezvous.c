/*
 * test_rendezvous.c -- CPU-only driver of host/mfuoco_rendezvous.c (compiled together with it; no GPU, no RCCL): one process per rank.
 *   usage: test_rendezvous <rank> <world> <id_file> <limit_s> <tag>      rank 0's id is 128 bytes of <tag> (one character)
 * prints "rank R id TT..TT" (the first 8 bytes of the id it ended up with) and exits 0, or exits 3 when the rendezvous gives up.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfuoco_rendezvous.h"

int main(int argc, char **argv)
{
  if (argc < 6) return 2;
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  uint8_t id[MFUOCO_RDV_ID_BYTES];
  memset(id, rank == 0 ? argv[5][0] : 0, sizeof id);
  if (mfuoco_rendezvous_files(rank, world, argv[3], id, atof(argv[4]))) return 3;
  printf("rank %d id ", rank);
  for (int i = 0; i < 8; i++) printf("%02x", id[i]);
  printf("\n");
  fflush(stdout);
  /* (the files are left in place: in the library ncclCommInitRank stands between the rendezvous and mfuoco_rendezvous_cleanup, so nobody removes a file
   * another rank still has to read; the test removes the directory) */
  return 0;
}
