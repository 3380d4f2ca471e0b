#!/bin/sh
# dev tool: rebuild ONE object of libmfhip.so with extra compiler flags and relink (timing variants of a kernel).
# usage: tools/build_variant.sh evalmm "-DMMS_SW=4 -DMMS_RQ=4 -mllvm -amdgpu-mfma-vgpr-form=1"      (tools/build_variant.sh evalmm "" restores)
set -e
cd "$(dirname "$0")/../c-lwe-snarks_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -I../include -Icsrc $2 -c -o build/$1.o csrc/$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libmfhip.so build/*.o
echo "rebuilt $1 with: $2"
