#!/bin/bash
# CPU sanitizer pass over the library's HOST code (GPU sanitizers are not available on the pool): the five translation
# units compiled with the HOST side instrumented (-fno-gpu-sanitize: device code as usual; without a device every compute entry point
# returns ORBX_ERR_NO_DEVICE) under AddressSanitizer + UndefinedBehaviorSanitizer, then the CPU tests that drive host
# logic -- the symbolic CSR / chunk tables / resident-CG chunk table (fem_plan), the frame-grid counting sort
# (orbm_sorted_frame), the extractor's frame planning (orbx_plan), argument checks, the no-device paths of the workspace pool, the FeatureVector co-iteration and
# the Fuse loop tails -- run against that build (ORBX_LIB).  usage: tools/asan_host.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
OUT=${ASAN_OUT:-/tmp/orbx_asan}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -fno-gpu-sanitize -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libasan"
for f in orbx_extract orbx_stereo orbm_match orbm_search fem; do
    $HIPCC $FLAGS -c orb_slam2_e_amd/csrc/$f.hip -o "$OUT/$f.o" &
done
wait
$HIPCC --offload-arch=gfx950 -fno-gpu-sanitize -shared -fPIC -fsanitize=address,undefined -shared-libasan -o "$OUT/liborbslam_hip.so" "$OUT"/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
echo "built $OUT/liborbslam_hip.so; runtime $RT"
ORBX_LIB="$OUT/liborbslam_hip.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_cpu_fem_plan.py tests/test_cpu_extract_plan.py tests/test_cpu_host_logic.py tests/test_fuse_and_projection.py tests/test_cpu_basics.py \
    -q -m "not gpu" -p no:cacheprovider "$@"
