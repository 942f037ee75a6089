#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code (the GPU pool offers no device sanitizer): the oracle through
# tests/test_oracle_cpu.py, and the host geometry / image-IO code through their check programs.  Builds into a scratch directory and
# puts the normal binaries back afterwards.  Run from the repo root (no GPU needed).
set -e
REPO=$(pwd); T=$(mktemp -d)
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -std=c++17 -o $T/geometry_check monocular_slam_amd/host/geometry_check.cpp monocular_slam_amd/host/Geometry.cpp
g++ $SAN -std=c++17 -o $T/imageio_check monocular_slam_amd/host/imageio_check.cpp monocular_slam_amd/host/ImageIO.cpp -lz
gcc $SAN -ffp-contract=off -shared -fPIC -o $T/liboracle.so oracle/*.c -lm
restore() { cp $T/geo.bak monocular_slam_amd/host/mslam_geometry_check; cp $T/img.bak monocular_slam_amd/host/mslam_imageio_check; cp $T/ora.bak oracle/liboracle.so; }
cp monocular_slam_amd/host/mslam_geometry_check $T/geo.bak; cp monocular_slam_amd/host/mslam_imageio_check $T/img.bak; cp oracle/liboracle.so $T/ora.bak
trap restore EXIT
cp $T/geometry_check monocular_slam_amd/host/mslam_geometry_check; cp $T/imageio_check monocular_slam_amd/host/mslam_imageio_check; cp $T/liboracle.so oracle/liboracle.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python3 -m pytest tests/test_host_geometry_cpu.py tests/test_frameloader_cpu.py -x -q
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python3 -m pytest tests/test_oracle_cpu.py -x -q
echo "sanitizers: clean"
