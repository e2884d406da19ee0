#!/bin/bash
# build_evidence.sh — what `__graft_entry__.build()` produces, measured: wall time of a CLEAN build of frieda_amd/csrc into a scratch
# directory (the in-tree library is left alone), the offload targets of the resulting code objects and the kernel count per target.
# No GPU needed (hipcc cross-compiles gfx950).  usage: bash tools/build_evidence.sh [out-file]   (default: stdout)
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/frieda_build_XXXXXX)
out=${1:-/dev/stdout}
{
  echo "# build evidence: $(date -u +%Y-%m-%dT%H:%M:%SZ), commit $(git -C $root rev-parse --short HEAD 2>/dev/null), $(/opt/rocm/bin/hipcc --version | grep -m1 -i 'HIP version')"
  t0=$(date +%s.%N)
  make -C $root/frieda_amd/csrc -s -j8 OUT=$tmp OBJ=$tmp/obj > $tmp/make.log 2>&1; rc=$?
  t1=$(date +%s.%N)
  echo "clean build: make -j8 rc=$rc, $(python3 -c "print(f'{$t1-$t0:.1f}')") s wall, $(ls $tmp/obj/*.o | wc -l) translation units, libfrieda_hip.so $(stat -c %s $tmp/libfrieda_hip.so) bytes, warnings: $(grep -c warning $tmp/make.log)"
  # the bundles inside the library (llvm-objdump --offloading extracts them next to its input: a scratch copy)
  ( cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading libfrieda_hip.so > offload.log 2>&1 )
  echo "offload bundles by target: $(grep -o 'bundle: .*' $tmp/offload.log | sed 's/.*\.so\.[0-9]*\.//' | sort | uniq -c | tr '\n' ';')"
  k=0
  for f in $tmp/libfrieda_hip.so.*.hipv4-amdgcn-amd-amdhsa--gfx950; do [ -f "$f" ] && k=$((k + $(/opt/rocm/lib/llvm/bin/llvm-readelf -s "$f" 2>/dev/null | grep -c '\.kd$'))); done
  echo "device kernels (kernel descriptors over all gfx950 code objects): $k"
  echo "code objects for any other target: $(ls $tmp | grep hipv4 | grep -vc gfx950)"
  echo "offload target of the Makefile: $(grep -m1 '^ARCH' $root/frieda_amd/csrc/Makefile)"
} > "$out"
rm -rf $tmp
