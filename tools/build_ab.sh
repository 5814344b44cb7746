#!/bin/bash
# A/B build: libmcedm_hip.so with some sources compiled differently -> m-cedm_amd/_ab/<name>.so (git-ignored; use with MCEDM_LIB=...)
#   tools/build_ab.sh <name> "<extra hipcc flags>" file1.hip [file2.hip ...]     (files relative to m-cedm_amd/csrc; or alternate paths)
set -e
name=$1; flags=$2; shift 2
R=/root/repo/m-cedm_amd; O=$R/_ab/$name; mkdir -p $O
objs=""
for f in $R/csrc/_build/*.o; do
  b=$(basename $f .o); skip=0
  for s in "$@"; do [ "$(basename $s .hip)" == "$b" ] && skip=1; done
  [ $skip == 0 ] && objs="$objs $f"
done
for s in "$@"; do
  src=$s; [ -f "$src" ] || src=$R/csrc/$s
  b=$(basename $s .hip)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I/root/repo/include -I$R/csrc -Wall -Wno-unused-function $flags -c $src -o $O/$b.o
  objs="$objs $O/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_ab/$name.so $objs
echo $R/_ab/$name.so
