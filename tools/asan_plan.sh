# the host planner (csrc/plan.cpp) on random batches under AddressSanitizer + UBSan, then ThreadSanitizer with a forced
# team of threads; CPU only:  bash tools/asan_plan.sh [batches] [seed]
set -eu
mkdir -p /tmp/lgmi_asan
F="-O1 -g -std=c++17 -fno-omit-frame-pointer -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
g++ $F -fsanitize=address,undefined -o /tmp/lgmi_asan/plan_fuzz tools/src/plan_fuzz.cpp l-giremi_amd/csrc/plan.cpp
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 /tmp/lgmi_asan/plan_fuzz "${1:-60}" "${2:-1}"
g++ $F -fsanitize=thread -o /tmp/lgmi_asan/plan_fuzz_tsan tools/src/plan_fuzz.cpp l-giremi_amd/csrc/plan.cpp
LGMI_PLAN_THREADS=6 /tmp/lgmi_asan/plan_fuzz_tsan "${1:-60}" "${2:-1}"
