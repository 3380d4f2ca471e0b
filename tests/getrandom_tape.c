/*
 * getrandom_tape.c -- TEST INFRASTRUCTURE: a getrandom(2) that serves a deterministic tape, for LD_PRELOAD into the reference's own test programs
 * (oracle/_ref/drivers/test_*), which draw every key, seed, message and error from the OS ("define getrandom in the
 * harness ... two runs give identical CRS and proof hashes").  The tape is the splitmix64 stream of $MF_TAPE_SEED (decimal or 0x...), 8 bytes per step,
 * a partial step's tail dropped -- tests/test_gpu_reference_drivers.py reproduces it in Python to know, before a program runs, which bytes it will see.
 * Without $MF_TAPE_SEED the call goes to the kernel.  Built by the test itself (gcc -shared -fPIC); never part of the product.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/types.h>
#include <unistd.h>

static int tape_state_known;
static int tape_on;
static uint64_t tape_state;

ssize_t getrandom(void *buf, size_t len, unsigned int flags)
{
  if (!tape_state_known) {
    const char *e = getenv("MF_TAPE_SEED");
    tape_on = e && *e;
    if (tape_on) tape_state = strtoull(e, NULL, 0);
    tape_state_known = 1;
  }
  if (!tape_on) return syscall(SYS_getrandom, buf, len, flags);
  uint8_t *p = buf;
  for (size_t i = 0; i < len; i += 8) {
    uint64_t z = (tape_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z ^= z >> 31;
    memcpy(p + i, &z, len - i < 8 ? len - i : 8);
  }
  return (ssize_t)len;
}
