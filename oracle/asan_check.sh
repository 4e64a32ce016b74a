#!/bin/bash
# One-off memory-safety check of the CPU oracle (test infrastructure) under ASan + UBSan.
# GPU sanitizers are not available on the pool; this covers the CPU build only.
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared -std=c99 \
    -o /tmp/liboracle_asan.so oracle/orb_oracle.c oracle/match_oracle.c oracle/fem_oracle.c oracle/stereo_oracle.c oracle/pose_nr_oracle.c -lm
cp oracle/liboracle.so /tmp/liboracle_backup.so
cp /tmp/liboracle_asan.so oracle/liboracle.so
touch oracle/liboracle.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_cpu_basics.py tests/test_cpu_fem.py tests/test_cpu_fem_two_level.py -x -q -k "not library_builds and not no_device and not cxx" || true
cp /tmp/liboracle_backup.so oracle/liboracle.so
touch oracle/liboracle.so
