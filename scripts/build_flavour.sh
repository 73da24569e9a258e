#!/bin/bash
# Build an experimental flavour of libbmf_hip.so that differs only in xf_bits_i8.hip: build_flavour.sh NAME "-DFLAG ..."
# -> pybmf_amd/csrc/libbmf_NAME.so (select it with BMF_LIB=libbmf_NAME.so).  The other objects come from build/.
set -e
cd "$(dirname "$0")/../pybmf_amd/csrc"
mkdir -p build_exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBMF_SHAPE16=1 -mllvm -amdgpu-mfma-vgpr-form=1 $2 -c ${3:-xf_bits_i8.hip} -o build_exp/$1.o
OBJS=$(ls build/*.o | grep -v "build/$(basename ${3:-xf_bits_i8.hip} .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libbmf_$1.so $OBJS build_exp/$1.o
echo built libbmf_$1.so
