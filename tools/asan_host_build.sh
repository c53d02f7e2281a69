#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libg16hip.so (device code unsanitised: GPU ASan is not available on this
# pool), for the CPU test-suite and the parser fuzzers:
#   bash tools/asan_host_build.sh            # -> /tmp/asan/libg16hip.so   (run from nzcp-circom_amd/csrc)
#   cp /tmp/asan/libg16hip.so nzcp-circom_amd/lib/ ; LD_PRELOAD=<clang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0 \
#       python -m pytest tests -m "not gpu"      (then restore the optimised library with make)
# r02: 61 host tests + 15 000 mutated zkey / r1cs / ptau inputs clean; the run found the unbounded allocations of the
# r1cs reader (header counts now checked against the section sizes).
cd "$(dirname "$0")/../nzcp-circom_amd/csrc"
mkdir -p /tmp/asan
set -e
HIPCC=/opt/rocm/bin/hipcc
OUT=/tmp/asan
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan"
for f in prover synth multi; do $HIPCC -O1 -g -std=c++17 -fPIC $SAN -c $f.cpp -o $OUT/$f.o & done
wait
for f in ntt qap msm_g2 ops setup_gpu verify plonk verify_plonk; do $HIPCC --offload-arch=gfx950 -O1 -std=c++17 -fPIC -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -c $f.hip -o $OUT/$f.o & done
wait
# msm_g1 is large: reuse the optimised, unsanitised object
cp ../build/msm_g1.o $OUT/msm_g1.o
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -o $OUT/libg16hip.so $OUT/*.o -lpthread
