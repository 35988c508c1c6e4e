#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: a side build of libapr_amd.so into tools/tmp_libs/libapr_NAME.so (A/B timing: APR_AMD_LIB=...)
set -e
cd "$(dirname "$0")/../apr_amd/csrc"
name=$1; shift
mkdir -p ../../tools/tmp_libs
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -pthread -Wall -Wno-unused-function -I ../../include "$@" \
  -o ../../tools/tmp_libs/libapr_$name.so engine.hip model_parse.cpp synth.cpp ingest.cpp
echo tools/tmp_libs/libapr_$name.so
