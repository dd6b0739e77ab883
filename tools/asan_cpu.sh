#!/bin/bash
# AddressSanitizer + UBSan over the host-side product code (probes, muxer, hashes, job front end) and the oracle, driven by the
# CPU test suite.  GPU sanitizers are not available on the pool, so the .hip objects are linked as built.  Run from the repo root
# after `make -C rawcooked_amd/csrc`; the regular libraries are put back afterwards.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd $R/rawcooked_amd/csrc
for f in rc_common formats mkv_mux hashes ffv1_host job; do
  g++ -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -I. -c $f.cpp -o $T/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $T/librcgpu.so $T/*.o build/ffv1_gpu.o build/ffv1_check.o build/flac_gpu.o build/pipeline.o -lpthread
cd $R/oracle
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $T/liboracle.so ffv1_oracle.c flac_oracle.c dpx_oracle.c -lm -lpthread
cd $R
cp rawcooked_amd/librcgpu.so $T/librcgpu.orig; cp oracle/liboracle.so $T/liboracle.orig
trap 'cp $T/librcgpu.orig rawcooked_amd/librcgpu.so; cp $T/liboracle.orig oracle/liboracle.so; rm -rf $T' EXIT
cp $T/librcgpu.so rawcooked_amd/librcgpu.so; cp $T/liboracle.so oracle/liboracle.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider ${ASAN_PYTEST_ARGS:-}
