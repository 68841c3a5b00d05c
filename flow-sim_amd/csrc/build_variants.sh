#!/bin/bash
# builds experiment variants of the library: name:flags
cd "$(dirname "$0")"
mkdir -p variants
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-slp-vectorize -ffp-contract=on -mllvm -enable-ipra=0 ${FS_MIN:--DFS_MINIMAL=1} "$@" -shared -o variants/lib_$name.so fs_abi.hip -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" ; }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags &
done
wait
ls -la variants/
