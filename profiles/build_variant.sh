#!/bin/bash
# A/B builds of libsvx.so: bash profiles/build_variant.sh <tag> "<extra compiler flags>" [only-this-source.hip]
#   ->  speech-vecalign_amd/svx/libsvx_<tag>.so   (select at run time with SVX_LIB=<path>; the product build stays
#       `make -C speech-vecalign_amd/csrc`).  With a third argument only that source gets the flags; the other objects are the tree's own.
set -e
TAG=$1; FLAGS=$2; ONLY=$3
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/svx_build_$TAG
rm -rf $B && mkdir -p $B/speech-vecalign_amd $B/include
cp -r $R/speech-vecalign_amd/csrc $B/speech-vecalign_amd/csrc
cp $R/include/svx.h $B/include/
mkdir -p $B/speech-vecalign_amd/svx
if [ -n "$ONLY" ]; then
  make -s -j8 -C $R/speech-vecalign_amd/csrc
  rm -f $B/speech-vecalign_amd/csrc/${ONLY%.hip}.o
  touch $B/speech-vecalign_amd/csrc/*.o
else
  rm -f $B/speech-vecalign_amd/csrc/*.o
fi
make -s -j8 -C $B/speech-vecalign_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS"
cp $B/speech-vecalign_amd/svx/libsvx.so $R/speech-vecalign_amd/svx/libsvx_$TAG.so
echo built $R/speech-vecalign_amd/svx/libsvx_$TAG.so
