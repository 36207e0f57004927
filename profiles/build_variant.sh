#!/bin/bash
# A/B builds of libsvx.so: bash profiles/build_variant.sh <tag> "<extra compiler flags>"  ->  speech-vecalign_amd/svx/libsvx_<tag>.so
# (select at run time with SVX_LIB=<path>; the product build stays `make -C speech-vecalign_amd/csrc`)
set -e
TAG=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/svx_build_$TAG
rm -rf $B && mkdir -p $B/speech-vecalign_amd $B/include
cp -r $R/speech-vecalign_amd/csrc $B/speech-vecalign_amd/csrc
cp $R/include/svx.h $B/include/
rm -f $B/speech-vecalign_amd/csrc/*.o
mkdir -p $B/speech-vecalign_amd/svx
make -s -j8 -C $B/speech-vecalign_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS"
cp $B/speech-vecalign_amd/svx/libsvx.so $R/speech-vecalign_amd/svx/libsvx_$TAG.so
echo built $R/speech-vecalign_amd/svx/libsvx_$TAG.so
