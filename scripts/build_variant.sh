#!/bin/bash
# A/B builds of the HIP library: scripts/build_variant.sh NAME [extra hipcc flags]  ->  difflinker_amd/variants/lib_NAME.so
# (select it with DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_NAME.so; *.so is git-ignored but travels to the GPU box with
# the snapshot - build/ does not)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/obj_$name difflinker_amd/variants
objs=""
for f in egnn_fc egnn_sparse size_gnn; do
  extra=""; [ $f = egnn_fc ] && extra="-mllvm -amdgpu-sched-strategy=iterative-ilp"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I include $extra "$@" \
      -c difflinker_amd/csrc/$f.hip -o build/obj_$name/$f.o &
  objs="$objs build/obj_$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o difflinker_amd/variants/lib_$name.so
rm -rf build/obj_$name
ls -la difflinker_amd/variants/lib_$name.so
