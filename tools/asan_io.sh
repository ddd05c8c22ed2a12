# liblgmi_io (host-only BAM reader + site extraction) under AddressSanitizer + UBSan, CPU only:  bash tools/asan_io.sh
set -eu
mkdir -p /tmp/lgmi_asan
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/lgmi_asan/liblgmi_io.so l-giremi_amd/csrc/bamio.cpp -lz -ldl
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 LGMI_IO_LIB=/tmp/lgmi_asan/liblgmi_io.so \
  python -m pytest tests/test_region_fast.py tests/test_bamio.py tests/test_cli.py -x -q -m "not gpu" -p no:cacheprovider
