#!/bin/bash
# tools/probe/build_variant.sh <name> <-D flags...>: libmi355clip.so with vit.hip compiled under other macros, for
# tools/tower_ab.py --lib tools/probe/variant/<name>/libmi355clip.so (A/B across processes on one box; git-ignored)
set -e
name=$1; shift
d=tools/probe/variant/$name; mkdir -p $d
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off "$@" -c image_search_amd/csrc/vit.hip -o $d/vit.o
c=image_search_amd/csrc
hipcc --offload-arch=gfx950 -shared -fPIC $c/core.o $c/knn.o $d/vit.o $c/preprocess.o $c/pipeline.o $c/sharded.o $c/index.o -ldl -o $d/libmi355clip.so
echo $d/libmi355clip.so
