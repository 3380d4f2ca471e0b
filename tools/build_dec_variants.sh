#!/bin/sh
# dev tool: builds of libmfhip.so that differ only in k_decrypt_mm's tile parameters, into tools/ab/ (git-ignored, travels to the GPU box):
#   tools/build_dec_variants.sh "1 4 16" "1 4 4" ...      (row tiles per wave, k-steps per group, waves per workgroup)
# then on the box:  for f in tools/ab/libmfhip_dec_*.so; do MFHIP_LIB=$PWD/$f python tools/decrypt_time.py 65536; done
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$R/tools/ab"
for v in "$@"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-result -Wno-unused-value -I"$R/include" -I"$R/c-lwe-snarks_amd/csrc" \
      -DDEC_RT=$1 -DDEC_GK=$2 -DDEC_WAVES=$3 $4 -c -o /tmp/encmm_variant.o "$R/c-lwe-snarks_amd/csrc/encmm.hip"
  B="$R/c-lwe-snarks_amd/build"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$R/tools/ab/libmfhip_dec_$1_$2_$3$4.so" $B/mfhip.o $B/poly.o $B/snark.o $B/evalmm.o /tmp/encmm_variant.o $B/expandmm.o
  echo "built $1 $2 $3 $4"
done
