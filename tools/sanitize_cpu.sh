#!/bin/bash
# ASan + UBSan pass over everything that runs on the CPU: the checker (oracle/*.c, through the
# whole "not gpu" suite) and the host-side C++ (TransferFunctions, VolumeFiles, through their
# drivers).  GPU sanitizers are not available on the pool, so this is the sanitizer coverage
# there is.  Leaves the normal builds in place when it ends.
set -e
cd "$(dirname "$0")/.."
H=simian-spacemonkey_amd/host
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -g"
restore() { make -s -C oracle clean all >/dev/null; python3 -c "import __graft_entry__ as g; g.build()" >/dev/null; }
trap restore EXIT
g++ -std=c++17 -O1 $SAN -I$H tests/host/files_main.cpp $H/VolumeFiles.cpp -o tests/host/files_main
g++ -std=c++17 -O1 $SAN -I$H tests/host/tf_main.cpp $H/TransferFunctions.cpp -o tests/host/tf_main
ASAN_OPTIONS=detect_leaks=1 python3 -m pytest tests/test_tf_frame.py tests/test_volume_files.py -x -q
(cd oracle && gcc -O1 -g -std=gnu99 -fPIC -fopenmp -ffp-contract=off -fno-fast-math $SAN -shared \
    -o liboracle.so smk_oracle.c smk_prep.c -lm)
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=4 python3 -m pytest tests -x -q -m "not gpu"
