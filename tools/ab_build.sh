#!/bin/bash
# A/B builds of libkmm.so: tools/ab_build.sh name1="-DFLAG ..." name2="..." -> build_ab/libkmm_<name>.so (in parallel).
# Run one with KMM_LIB_PATH=build_ab/libkmm_<name>.so; tools/ab_run.py benches them all on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
pids=()
for spec in "$@"; do
    name="${spec%%=*}"; flags="${spec#*=}"
    [ "$name" = "$spec" ] && flags=""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude $flags \
        -o "build_ab/libkmm_${name}.so" kmer_mapper_amd/csrc/kmm.hip 2> "build_ab/${name}.log" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
ls -la build_ab/*.so
exit $rc
