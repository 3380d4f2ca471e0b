#!/bin/sh
# dev tool: builds of libmfhip.so that differ only in evalmm.hip's compile flags, into tools/ab/ (git-ignored, travels to the GPU box):
#   tools/build_mm_variants.sh base "" prio_young "-DMMS_PRIO_YOUNG" ...     then on the box:  python tools/ab_rounds.py tools/ab/libmfhip_mm_*.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$R/tools/ab"
B="$R/c-lwe-snarks_amd/build"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-result -Wno-unused-value -I"$R/include" -I"$R/c-lwe-snarks_amd/csrc" $flags -c -o /tmp/evalmm_variant.o "$R/c-lwe-snarks_amd/csrc/evalmm.hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$R/tools/ab/libmfhip_mm_$name.so" $B/mfhip.o $B/poly.o $B/snark.o /tmp/evalmm_variant.o $B/encmm.o $B/expandmm.o
  echo "built $name: $flags"
done
